// readers.hip -- Julius' BINARY model files read directly, host code.
//
//   jamd_binhmm_to_blob() / jamd_gmm_load_binhmm()   the binary HMM definition mkbinhmm writes
//       (libsent/src/hmminfo/write_binhmm.c, read back by read_binhmm.c:756-903) -> the flat model of
//       jamd_gmm_desc / the "JAMDGMM1" blob, byte for byte what jamd_export produces from the same file
//       through Julius' own reader + julius_amd/shim/jamd_flatten.c (tests/test_readers.py compares the files).
// Not covered: the tree lexicon and its LM tables.  The lexicon is not a file format but the output of
// libjulius/src/wchmm.c (build_wchmm2(): 2 000 lines of tree building, cross-word context handling and factoring
// set-up over the dictionary, the HMMList AND the LM -- the factoring values and the words kept out of the tree are
// functions of the 1-gram); PREFIX.lex, tree and N-gram tables together, therefore comes from jamd_export, which links
// Julius' own loaders.  gzip-compressed files are refused with a message.
#include "jamd_internal.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Reader {
  FILE *f = nullptr;
  bool ok = true, swap = true;          // binhmm: always big-endian on disk (read_binhmm.c:69-73)
  std::string path;
  ~Reader() { if (f) fclose(f); }
  bool open(const char *p) {
    path = p;
    f = fopen(p, "rb");
    if (!f) { jamd_set_error("cannot open %s", p); return false; }
    unsigned char m[2] = {0, 0};
    if (fread(m, 1, 2, f) == 2 && m[0] == 0x1f && m[1] == 0x8b) { jamd_set_error("%s: gzip-compressed; decompress it first", p); return false; }
    rewind(f);
    return true;
  }
  void raw(void *buf, size_t unit, size_t n) {
    if (!ok) return;
    if (n && fread(buf, unit, n, f) != n) { ok = false; jamd_set_error("%s: unexpected end of file", path.c_str()); return; }
    if (swap && unit > 1) {
      unsigned char *b = (unsigned char *)buf;
      for (size_t i = 0; i < n; i++, b += unit)
        for (size_t k = 0; k < unit / 2; k++) { unsigned char t = b[k]; b[k] = b[unit - 1 - k]; b[unit - 1 - k] = t; }
    }
  }
  template <typename T> T get() { T v{}; raw(&v, sizeof(T), 1); return v; }
  std::string str() {                    // rdn_str(): bytes up to and including NUL
    std::string s;
    int c;
    while (ok && (c = fgetc(f)) != EOF) { if (c == 0) return s; s.push_back((char)c); if (s.size() > 65536) break; }
    ok = false; jamd_set_error("%s: unterminated string", path.c_str());
    return s;
  }
  void skip(size_t bytes) { if (ok && bytes && fseek(f, (long)bytes, SEEK_CUR) != 0) ok = false; }
};

struct Pdf { bool tmix = false; int book = -1; std::vector<unsigned> dens; std::vector<float> w; };

struct FlatGmm {
  int S = 0, D = 0, G = 0, E = 0, nbook = 0;
  std::vector<float> mean, ivar, gconst, ent_logw;
  std::vector<int> st_off, ent_dens, st_book;
};

// rd_pdf_sub(), read_binhmm.c:544-585
bool read_pdf(Reader &r, Pdf &p, const std::vector<std::vector<unsigned>> &books) {
  short n = r.get<short>();
  if (!r.ok) return false;
  if (n == -1) {
    const unsigned b = r.get<unsigned>();
    if (!r.ok || b >= books.size()) { jamd_set_error("%s: codebook id out of range", r.path.c_str()); return false; }
    p.tmix = true; p.book = (int)b; p.dens = books[b];
  } else {
    if (n < 0) { jamd_set_error("%s: negative mixture count", r.path.c_str()); return false; }
    p.dens.resize((size_t)n);
    r.raw(p.dens.data(), 4, (size_t)n);
  }
  p.w.resize(p.dens.size());
  r.raw(p.w.data(), 4, p.w.size());
  return r.ok;
}

bool read_binhmm(const char *path, FlatGmm &out) {
  Reader r;
  if (!r.open(path)) return false;
  // rd_header(), read_binhmm.c:193-257
  const std::string h = r.str();
  bool embed = false, inversed = false, mpdf_macro = false;
  if (h == "JBINHMM\n") {
  } else if (h == "JBINHMMV2") {
    const std::string q = r.str();
    for (size_t i = 0; i + 1 < q.size() + 1 && i < q.size(); i += 2) {
      if (q[i] != '_' || i + 1 >= q.size()) break;
      if (q[i + 1] == 'P') embed = true;
      else if (q[i + 1] == 'V') inversed = true;
      else if (q[i + 1] == 'M') mpdf_macro = true;
      else { jamd_set_error("%s: unknown header qualifier '%c'", path, q[i + 1]); return false; }
    }
  } else { jamd_set_error("%s: not a Julius binary HMM file", path); return false; }
  if (embed) {                           // rd_para(), :125-179: skipped, the scoring path does not use it
    const short ver = r.get<short>();
    int a[4];
    r.raw(a, 4, 4);
    if (a[0] == 0 && a[2] == 0 && a[1] != 0 && a[3] != 0) r.skip(8);      // files written on 64-bit hosts by old versions
    r.skip(4 + 4 * 4 + 4 * 2 + 4 * 4);  // preEmph, lifter..accWin, silFloor escale, hipass..raw_e
    if (ver == 1) r.skip(8);
    r.skip(4);                          // zmeanframe
    if (ver >= 3) r.skip(4);            // usepower
  }
  // rd_opt() :261-271, rd_type() :280-285
  const short nstream = r.get<short>();
  r.skip(2 * 50);                       // stream_info.vsize[MAXSTREAMNUM]
  const short vec_size = r.get<short>();
  r.skip(2 * 3);                        // cov_type, dur_type, param_type
  const unsigned char tied = r.get<unsigned char>();
  r.skip(4);                            // maxmixturenum
  if (!r.ok) return false;
  if (nstream != 1) { jamd_set_error("%s: %d streams (the device path serves single-stream models)", path, (int)nstream); return false; }
  const int D = vec_size;
  // rd_trans() :303-341: not part of the state pool
  const unsigned ntr = r.get<unsigned>();
  for (unsigned i = 0; r.ok && i < ntr; i++) { r.str(); const short n = r.get<short>(); if (n < 0) { r.ok = false; break; } r.skip((size_t)n * n * 4); }
  // rd_var() :352-378
  const unsigned nvar = r.get<unsigned>();
  if (!r.ok || nvar > (1u << 28)) { jamd_set_error("%s: bad variance count", path); return false; }
  std::vector<std::vector<float>> var(nvar);
  for (unsigned i = 0; r.ok && i < nvar; i++) {
    r.str();
    const short len = r.get<short>();
    if (len < 0) { r.ok = false; break; }
    var[i].resize((size_t)len);
    r.raw(var[i].data(), 4, (size_t)len);
    if (!inversed)                      // htk_hmm_inverse_variances(), rdhmmdef.c:162-172: 1.0 / v in double
      for (float &x : var[i]) x = (float)(1.0 / (double)x);
  }
  // rd_dens() :396-427
  const unsigned ndens = r.get<unsigned>();
  if (!r.ok || ndens > (1u << 28)) { jamd_set_error("%s: bad density count", path); return false; }
  struct Dens { std::vector<float> mean; unsigned var; float gconst; };
  std::vector<Dens> dens(ndens);
  for (unsigned i = 0; r.ok && i < ndens; i++) {
    r.str();
    const short len = r.get<short>();
    if (len < 0) { r.ok = false; break; }
    dens[i].mean.resize((size_t)len);
    r.raw(dens[i].mean.data(), 4, (size_t)len);
    dens[i].var = r.get<unsigned>();
    dens[i].gconst = r.get<float>();
    if (r.ok && dens[i].var >= nvar) { jamd_set_error("%s: variance id out of range", path); return false; }
  }
  // rd_tmix() :489-527
  std::vector<std::vector<unsigned>> books;
  if (tied) {
    const unsigned nb = r.get<unsigned>();
    if (!r.ok || nb > (1u << 24)) { jamd_set_error("%s: bad codebook count", path); return false; }
    books.resize(nb);
    for (unsigned i = 0; r.ok && i < nb; i++) {
      r.str();
      const int n = r.get<int>();
      if (n < 0) { r.ok = false; break; }
      books[i].resize((size_t)n);
      r.raw(books[i].data(), 4, (size_t)n);
    }
  }
  // rd_mpdf() :588-614
  std::vector<Pdf> mpdf;
  if (mpdf_macro) {
    const unsigned n = r.get<unsigned>();
    if (!r.ok || n > (1u << 28)) { jamd_set_error("%s: bad mixture pdf count", path); return false; }
    mpdf.resize(n);
    for (unsigned i = 0; r.ok && i < n; i++) { r.str(); r.skip(2); if (!read_pdf(r, mpdf[i], books)) return false; }
  }
  // rd_state() :633-697: a state's id is its index in the file
  const unsigned nst = r.get<unsigned>();
  if (!r.ok || nst == 0 || nst > (1u << 28)) { jamd_set_error("%s: bad state count", path); return false; }
  std::vector<Pdf> st(nst);
  for (unsigned i = 0; r.ok && i < nst; i++) {
    r.str();
    if (mpdf_macro) {
      const unsigned mid = r.get<unsigned>();
      if (!r.ok || mid >= mpdf.size()) { jamd_set_error("%s: state %u without a mixture pdf", path, i); return false; }
      st[i] = mpdf[mid];
    } else if (!read_pdf(r, st[i], books)) return false;
  }
  if (!r.ok) return false;

  // the walk of julius_amd/shim/jamd_flatten.c over hmminfo->ststart: state_add() prepends (rdhmmdef_state.c:67-68),
  // so the list runs from the LAST state of the file to the first; densities are numbered at first sight
  out.S = (int)nst; out.D = D; out.nbook = tied ? (int)books.size() : 0;
  out.st_off.assign(nst + 1, 0); out.st_book.assign(nst, -1);
  for (unsigned s = 0; s < nst; s++) out.st_off[s + 1] = out.st_off[s] + (int)st[s].dens.size();
  out.E = out.st_off[nst];
  out.ent_dens.assign((size_t)(out.E ? out.E : 1), -1); out.ent_logw.assign((size_t)(out.E ? out.E : 1), 0.0f);
  std::vector<int> gid(ndens, -1);
  for (int s = (int)nst - 1; s >= 0; s--) {
    const Pdf &p = st[(size_t)s];
    out.st_book[(size_t)s] = p.tmix ? p.book : -1;
    for (size_t i = 0; i < p.dens.size(); i++) {
      const unsigned d = p.dens[i];
      int g = -1;
      if (d < ndens) {                   // an id >= dens_num stands for a NULL density (:515-519, :563-567)
        if ((int)dens[d].mean.size() != D || (int)var[dens[d].var].size() != D) { jamd_set_error("%s: density %u is not %d-dimensional", path, d, D); return false; }
        if (gid[d] < 0) {
          gid[d] = out.G++;
          out.mean.insert(out.mean.end(), dens[d].mean.begin(), dens[d].mean.end());
          out.ivar.insert(out.ivar.end(), var[dens[d].var].begin(), var[dens[d].var].end());
          out.gconst.push_back(dens[d].gconst);
        }
        g = gid[d];
      }
      out.ent_dens[(size_t)out.st_off[(size_t)s] + i] = g;
      out.ent_logw[(size_t)out.st_off[(size_t)s] + i] = p.w[i];
    }
  }
  if (out.G == 0) { jamd_set_error("%s: no Gaussian density is referenced", path); return false; }
  return true;
}

// ---- blob writer (same container as julius_amd/shim/jamd_flatten.c gmm_put()) ---------------------------
bool put(FILE *f, const char *name, int dtype, int count, const void *data) {
  char nm[24];
  memset(nm, 0, sizeof(nm)); strncpy(nm, name, sizeof(nm) - 1);
  if (fwrite(nm, 1, 24, f) != 24 || fwrite(&dtype, 4, 1, f) != 1 || fwrite(&count, 4, 1, f) != 1) return false;
  if (dtype == 2) {
    static const char zero[4] = {0, 0, 0, 0};
    if (count > 0 && fwrite(data, 1, (size_t)count, f) != (size_t)count) return false;
    if ((count & 3) && fwrite(zero, 1, (size_t)(4 - (count & 3)), f) != (size_t)(4 - (count & 3))) return false;
    return true;
  }
  return count <= 0 || fwrite(data, 4, (size_t)count, f) == (size_t)count;
}

// ---- binary N-gram ---------------------------------------------------------------------------------------
struct Tuple {                            // NGRAM_TUPLE_INFO, libsent/include/sent/ngram2.h:137-156
  unsigned totalnum = 0, bgnlistlen = 0, context_num = 0;
  bool is24bit = false, ct_compaction = false;
  std::vector<unsigned> bgn; std::vector<unsigned> num; std::vector<unsigned> nnid2wid;
  std::vector<float> prob, bo_wt;
};

bool read_bingram(const char *path, int &n_out, int &dir_out, bool &reversed, std::vector<std::string> &wname,
                  std::vector<Tuple> &t, std::vector<float> &bo_wt_1, std::vector<float> &p_2) {
  Reader r;
  if (!r.open(path)) return false;
  char hd[512];
  r.swap = false;
  r.raw(hd, 1, 512);
  if (!r.ok) return false;
  hd[511] = 0;
  if (strncmp(hd, "julius_bingram_v5", 17) != 0) {
    jamd_set_error("%s: not a julius_bingram_v5 file (older versions are converted by mkbingram)", path); return false;
  }
  // second header line: "word=<size> byteorder=LE|BE" (ngram_read_bin.c check_header())
  const char *l2 = strchr(hd, '\n');
  if (!l2 || !strstr(l2, "word=4byte(int)")) { jamd_set_error("%s: 2-byte word ids (WORDS_INT build expected)", path); return false; }
  const char *bo = strstr(l2, "byteorder=");
  r.swap = bo != nullptr && strncmp(bo + 10, "BE", 2) == 0;          // files of this version carry their writer's order
  if (!bo) r.swap = true;                                            // no tag: big-endian (older writers)
  const int n = r.get<int>(), dir = r.get<int>();
  const unsigned char rev = r.get<unsigned char>();
  if (!r.ok || n < 2 || n > 10) { jamd_set_error("%s: N=%d", path, n); return false; }
  n_out = n; dir_out = dir; reversed = rev != 0;
  t.assign((size_t)n, Tuple());
  for (int m = 0; m < n; m++) t[(size_t)m].totalnum = r.get<unsigned>();
  const int wlen = r.get<int>();
  if (!r.ok || wlen < 0) { jamd_set_error("%s: bad word list", path); return false; }
  std::vector<char> names((size_t)wlen + 1, 0);
  r.raw(names.data(), 1, (size_t)wlen);
  for (int p = 0; p < wlen;) { wname.emplace_back(names.data() + p); p += (int)wname.back().size() + 1; }
  if (!r.ok || wname.size() != t[0].totalnum) { jamd_set_error("%s: %zu names for %u words", path, wname.size(), t[0].totalnum); return false; }
  for (int m = 0; m < n && r.ok; m++) {
    Tuple &x = t[(size_t)m];
    x.is24bit = r.get<unsigned char>() != 0; x.ct_compaction = r.get<unsigned char>() != 0;
    x.bgnlistlen = r.get<unsigned>(); x.context_num = r.get<unsigned>();
    if (!r.ok || x.totalnum > (1u << 30) || x.bgnlistlen > (1u << 30) || x.context_num > (1u << 30)) { r.ok = false; break; }
    if (m > 0) {
      x.bgn.resize(x.bgnlistlen);
      if (x.is24bit) {
        std::vector<unsigned char> up(x.bgnlistlen); std::vector<unsigned short> lo(x.bgnlistlen);
        r.raw(up.data(), 1, up.size()); r.raw(lo.data(), 2, lo.size());
        for (size_t i = 0; i < up.size(); i++) x.bgn[i] = up[i] == 255 ? 0xffffffffu : ((unsigned)up[i] << 16) | lo[i];   // NNID_INVALID_UPPER
      } else r.raw(x.bgn.data(), 4, x.bgn.size());
      x.num.resize(x.bgnlistlen);
      r.raw(x.num.data(), 4, x.num.size());                          // WORD_ID = int in the WORDS_INT build
      x.nnid2wid.resize(x.totalnum);
      r.raw(x.nnid2wid.data(), 4, x.nnid2wid.size());
    }
    x.prob.resize(x.totalnum);
    r.raw(x.prob.data(), 4, x.prob.size());
    if (r.get<int>() != 0) { x.bo_wt.resize(x.context_num); r.raw(x.bo_wt.data(), 4, x.bo_wt.size()); }
    if (r.get<int>() != 0) r.skip((size_t)x.totalnum * 3);           // nnid2ctid (only for N >= 3 lookups)
  }
  if (r.ok && r.get<int>() != 0) { bo_wt_1.resize(t[0].context_num); r.raw(bo_wt_1.data(), 4, bo_wt_1.size()); }
  if (r.ok && r.get<int>() != 0) { p_2.resize(t[1].totalnum); r.raw(p_2.data(), 4, p_2.size()); }
  if (!r.ok) { jamd_set_error("%s: truncated or malformed", path); return false; }
  return true;
}

}  // namespace

// The tables the first pass reads from a binary N-gram (which 2-gram: bi_prob_func_set(), ngram_access.c:449-466;
// DIR_LR = 0, DIR_RL = 1), in the arrays of jamd_lexicon_desc.
bool jamd_read_bingram_tables(const char *path, JamdNgramTables &o) {
  int n = 0, dir = 0; bool reversed = false;
  std::vector<std::string> wname; std::vector<Tuple> t; std::vector<float> bo_wt_1, p_2;
  if (!read_bingram(path, n, dir, reversed, wname, t, bo_wt_1, p_2)) return false;
  const Tuple &t1 = t[0], &t2 = t[1];
  const int V = (int)t1.totalnum;
  const std::vector<float> *bo, *bp;
  if (reversed) { o.mode = JAMD_NG_ADDITIONAL_OLD; bo = &bo_wt_1; bp = &p_2; }
  else if (dir == 0) { o.mode = JAMD_NG_NORMAL; bo = &t1.bo_wt; bp = &t2.prob; }
  else if (!bo_wt_1.empty()) { o.mode = JAMD_NG_ADDITIONAL; bo = &bo_wt_1; bp = &p_2; }
  else { o.mode = JAMD_NG_COMPUTE; bo = &t1.bo_wt; bp = &t2.prob; }
  if ((int)bo->size() < V || bp->size() < t2.totalnum || (int)t2.bgn.size() < V || (int)t2.num.size() < V || (int)t1.prob.size() < V) {
    jamd_set_error("%s: 2-gram tables shorter than the vocabulary", path); return false;
  }
  o.nword = V; o.nbigram = (int)t2.totalnum; o.n = n; o.dir = dir;
  o.uni_prob.assign(t1.prob.begin(), t1.prob.begin() + V);
  o.uni_bo.assign(bo->begin(), bo->begin() + V);
  o.bi_prob.assign(bp->begin(), bp->begin() + t2.totalnum);
  o.bi_bgn.resize((size_t)V); o.bi_num.resize((size_t)V); o.bi_wid.resize(t2.totalnum);
  for (int i = 0; i < V; i++) { o.bi_bgn[(size_t)i] = t2.bgn[(size_t)i] == 0xffffffffu ? -1 : (int)t2.bgn[(size_t)i]; o.bi_num[(size_t)i] = (int)t2.num[(size_t)i]; }
  for (size_t i = 0; i < o.bi_wid.size(); i++) o.bi_wid[i] = (int)t2.nnid2wid[i];
  o.names.clear();
  for (const std::string &w : wname) { o.names += w; o.names.push_back('\0'); }
  return true;
}

extern "C" {

int jamd_binhmm_to_blob(const char *binhmm_path, const char *blob_path) {
  if (!binhmm_path || !blob_path) { jamd_set_error("jamd_binhmm_to_blob: NULL argument"); return JAMD_EINVAL; }
  FlatGmm g;
  if (!read_binhmm(binhmm_path, g)) return JAMD_EINVAL;
  FILE *f = fopen(blob_path, "wb");
  if (!f) { jamd_set_error("cannot write %s", blob_path); return JAMD_EINVAL; }
  const int nrec = 8, ints[6] = {g.S, g.D, g.G, g.E, g.nbook, 1};
  bool ok = fwrite("JAMDGMM1", 1, 8, f) == 8 && fwrite(&nrec, 4, 1, f) == 1;
  ok = ok && put(f, "ints", 0, 6, ints) && put(f, "mean", 1, g.G * g.D, g.mean.data()) && put(f, "ivar", 1, g.G * g.D, g.ivar.data()) &&
       put(f, "gconst", 1, g.G, g.gconst.data()) && put(f, "st_off", 0, g.S + 1, g.st_off.data()) &&
       put(f, "ent_dens", 0, g.E, g.ent_dens.data()) && put(f, "ent_logw", 1, g.E, g.ent_logw.data()) &&
       put(f, "st_book", 0, g.S, g.st_book.data());
  if (fclose(f) != 0) ok = false;
  if (!ok) { jamd_set_error("write error on %s", blob_path); return JAMD_EINVAL; }
  return JAMD_OK;
}

int jamd_gmm_load_binhmm(jamd_engine *e, const char *binhmm_path, int gprune, int gprune_num, jamd_gmm **out) {
  if (!e || !binhmm_path || !out) { jamd_set_error("jamd_gmm_load_binhmm: NULL argument"); return JAMD_EINVAL; }
  *out = nullptr;
  FlatGmm g;
  if (!read_binhmm(binhmm_path, g)) return JAMD_EINVAL;
  jamd_gmm_desc d;
  memset(&d, 0, sizeof(d));
  d.nstate = g.S; d.veclen = g.D; d.ndens = g.G; d.nentry = g.E; d.nbook = g.nbook; d.nstream = 1;
  d.mean = g.mean.data(); d.ivar = g.ivar.data(); d.gconst = g.gconst.data(); d.st_off = g.st_off.data();
  d.ent_dens = g.ent_dens.data(); d.ent_logw = g.ent_logw.data(); d.st_book = g.st_book.data();
  return jamd_gmm_create(e, &d, gprune, gprune_num, out);
}

}  // extern "C"
