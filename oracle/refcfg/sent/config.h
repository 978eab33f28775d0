/* Hand-written build configuration for compiling the UNMODIFIED reference
 * libsent sources (read in place from $JULIUS_REF, default /root/reference)
 * into oracle/_ref/.  This replaces what the reference's ./configure would
 * generate; it selects the same "fast"/words-int feature set the survey
 * build used minus audio devices (no USE_MIC).
 * Test infrastructure only -- nothing in the product links against it. */
#ifndef JAMD_REFCFG_SENT_CONFIG_H
#define JAMD_REFCFG_SENT_CONFIG_H
#define LIBSENT_VERSION "4.6"
#define AUDIO_API_NAME "none"
#define AUDIO_API_DESC "no audio device (oracle build)"
#define AUDIO_FORMAT_DESC "RAW and WAV only"
#define GZIP_READING_DESC "zlib library"
#define STDC_HEADERS 1
#define WORDS_INT 1
#define USE_ADDLOG_ARRAY 1
#define HAVE_SOCKLEN_T 1
#define HAVE_UNISTD_H 1
#define HAVE_ZLIB 1
#define HAVE_STRCASECMP 1
#define HAVE_SLEEP 1
#define CLASS_NGRAM 1
#define MFCC_SINCOS_TABLE 1
#define USE_MBR 1
#define HAS_SIMD_FMA 1
#define HAS_SIMD_AVX 1
#define HAS_SIMD_SSE 1
#endif
