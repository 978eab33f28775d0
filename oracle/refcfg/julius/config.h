/* Hand-written build configuration for compiling the UNMODIFIED reference
 * libjulius sources into oracle/_ref/ ("fast" setup, the reference default:
 * UNIGRAM_FACTORING, PASS1_IWCD, SCAN_BEAM, GPRUNE_DEFAULT_BEAM...).
 * Test infrastructure only. */
#ifndef JAMD_REFCFG_JULIUS_CONFIG_H
#define JAMD_REFCFG_JULIUS_CONFIG_H
#define JULIUS_PRODUCTNAME "JuliusLib"
#define JULIUS_VERSION "4.6"
#define JULIUS_SETUP "fast"
#define JULIUS_HOSTINFO "x86_64-unknown-linux-gnu"
#define JULIUS_BUILD_INFO "oracle/_ref build"
#define RETSIGTYPE void
#define STDC_HEADERS 1
#define HAVE_PTHREAD 1
#define UNIGRAM_FACTORING 1
#define LOWMEM2 1
#define PASS1_IWCD 1
#define SCAN_BEAM 1
#define GPRUNE_DEFAULT_BEAM 1
#define CONFIDENCE_MEASURE 1
#define LM_FIX_DOUBLE_SCORING 1
#define GRAPHOUT_DYNAMIC 1
#define GRAPHOUT_SEARCH 1
#define ENABLE_PLUGIN 1
#endif
