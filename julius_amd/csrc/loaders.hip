// loaders.hip -- on-disk formats -> device models, without a Julius process.
//
//   jamd_gmm_load()      "JAMDGMM1" blob  (written by jamd_gmm_save(), julius_amd/shim/jamd_flatten.c,
//                                          from the HTK_HMM_INFO Julius' own hmmdefs / binhmm reader built)
//   jamd_gms_load()      "JAMDGMM1" blob with the records "state2gs" and "gms" (jamd_gms_save(): the
//                                          selection model of -gshmm and its state map)
//   jamd_lexicon_load()  "JAMDLEX1" blob  (written by jamd_lexicon_save(), julius_amd/shim/jamd_flatten_lex.c,
//                                          from the tree lexicon + LM tables of a RecogProcess)
//   jamd_dnn_load()      Julius' own DNN definition: the -dnnconf text file, the NumPy .npy weight and
//                        bias files it names and the state prior list -- read as the reference does
//                        (libjulius/src/m_jconf.c:577-735 dnn_config_file_parse(),
//                         libsent/src/phmm/calc_dnn.c:225-335 load_npy(), :390-434 dnn_layer_load(),
//                         :678-707 prior file), so the model equals DNNData after dnn_setup().
// Blob layout: magic[8], int32 nrec, then per record char name[24], int32 dtype (0 int32, 1 float32,
// 2 uint8), int32 count, payload padded to 4 bytes; scalars travel in the records "ints"/"floats".
// Host-only code; the kernels are in the other translation units.
#include "jamd_internal.h"
#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

struct Rec { int dtype = 0, count = 0; std::vector<unsigned char> data; };
using Blob = std::map<std::string, Rec>;

bool read_blob(const char *path, const char *magic, Blob &out) {
  FILE *f = fopen(path, "rb");
  if (!f) { jamd_set_error("cannot open %s", path); return false; }
  char m[8]; int nrec = 0;
  bool ok = fread(m, 1, 8, f) == 8 && memcmp(m, magic, 8) == 0 && fread(&nrec, 4, 1, f) == 1 && nrec > 0 && nrec < 4096;
  if (!ok) jamd_set_error("%s: not a %.8s file", path, magic);
  for (int i = 0; ok && i < nrec; i++) {
    char name[25]; Rec r;
    name[24] = 0;
    ok = fread(name, 1, 24, f) == 24 && fread(&r.dtype, 4, 1, f) == 1 && fread(&r.count, 4, 1, f) == 1 &&
         r.dtype >= 0 && r.dtype <= 2 && r.count >= 0;
    if (!ok) break;
    const size_t bytes = (size_t)r.count * (r.dtype == 2 ? 1 : 4), padded = (bytes + 3) & ~(size_t)3;
    r.data.resize(padded ? padded : 4);
    ok = padded == 0 || fread(r.data.data(), 1, padded, f) == padded;
    if (ok) out[name] = std::move(r);
  }
  if (!ok && nrec > 0) jamd_set_error("%s: truncated or corrupt record table", path);
  fclose(f);
  return ok;
}

// typed view of a record; count < 0 accepts any length
template <typename T>
const T *view(const Blob &b, const char *name, int dtype, long long count, bool &ok) {
  auto it = b.find(name);
  if (it == b.end() || it->second.dtype != dtype || (count >= 0 && it->second.count != count)) {
    if (ok) jamd_set_error("blob record \"%s\" missing or of unexpected type/size", name);
    ok = false;
    return nullptr;
  }
  return reinterpret_cast<const T *>(it->second.data.data());
}

// ---- NumPy .npy, as load_npy() reads it (calc_dnn.c:225-335): magic, version 1.0, little-endian
// float32 ('<f4'), two-dimensional (or (n,) / (n,1) for a bias), C order; the payload is taken raw
bool read_npy(const std::string &path, long long want, std::vector<float> &out) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) { jamd_set_error("cannot open %s", path.c_str()); return false; }
  unsigned char h[10];
  bool ok = fread(h, 1, 10, f) == 10 && memcmp(h, "\x93NUMPY", 6) == 0 && h[6] == 1;
  if (!ok) { jamd_set_error("%s: not a version-1 .npy file", path.c_str()); fclose(f); return false; }
  const int hlen = h[8] | (h[9] << 8);
  std::string hdr(hlen, '\0');
  ok = fread(&hdr[0], 1, hlen, f) == (size_t)hlen;
  if (ok && hdr.find("'<f4'") == std::string::npos) { jamd_set_error("%s: dtype is not '<f4'", path.c_str()); ok = false; }
  long long n = 1;
  if (ok) {                                           // 'shape': (a, b) -- product must match
    const size_t p = hdr.find("'shape'"), l = hdr.find('(', p), r = hdr.find(')', l);
    if (p == std::string::npos || l == std::string::npos || r == std::string::npos) ok = false;
    else {
      const char *c = hdr.c_str() + l + 1;
      while (c < hdr.c_str() + r) {
        char *e = nullptr;
        const long long v = strtoll(c, &e, 10);
        if (e == c) { c++; continue; }
        n *= v; c = e;
      }
    }
    if (!ok || n != want) { jamd_set_error("%s: %lld values, expected %lld", path.c_str(), n, want); ok = false; }
  }
  if (ok) { out.resize((size_t)want); ok = fread(out.data(), 4, (size_t)want, f) == (size_t)want; if (!ok) jamd_set_error("%s: short read", path.c_str()); }
  fclose(f);
  return ok;
}

std::string dir_of(const std::string &p) { const size_t s = p.rfind('/'); return s == std::string::npos ? "" : p.substr(0, s + 1); }
std::string rel(const std::string &dir, const std::string &v) { return (!v.empty() && v[0] == '/') ? v : dir + v; }   // filepath(), m_jconf.c

}  // namespace

extern "C" {

int jamd_gmm_load(jamd_engine *e, const char *path, int gprune, int gprune_num, jamd_gmm **out) {
  if (!e || !path || !out) { jamd_set_error("jamd_gmm_load: NULL argument"); return JAMD_EINVAL; }
  *out = nullptr;
  Blob b;
  if (!read_blob(path, "JAMDGMM1", b)) return JAMD_EINVAL;
  bool ok = true;
  const int *ints = view<int>(b, "ints", 0, 6, ok);
  if (!ok) return JAMD_EINVAL;
  jamd_gmm_desc d;
  memset(&d, 0, sizeof(d));
  d.nstate = ints[0]; d.veclen = ints[1]; d.ndens = ints[2]; d.nentry = ints[3]; d.nbook = ints[4]; d.nstream = ints[5];
  if (d.nstate <= 0 || d.veclen <= 0 || d.ndens <= 0 || d.nentry <= 0) { jamd_set_error("%s: bad sizes", path); return JAMD_EINVAL; }
  d.mean = view<float>(b, "mean", 1, (long long)d.ndens * d.veclen, ok);
  d.ivar = view<float>(b, "ivar", 1, (long long)d.ndens * d.veclen, ok);
  d.gconst = view<float>(b, "gconst", 1, d.ndens, ok);
  d.st_off = view<int>(b, "st_off", 0, d.nstate + 1, ok);
  d.ent_dens = view<int>(b, "ent_dens", 0, d.nentry, ok);
  d.ent_logw = view<float>(b, "ent_logw", 1, d.nentry, ok);
  if (b.count("st_book")) d.st_book = view<int>(b, "st_book", 0, d.nstate, ok);
  if (!ok) return JAMD_EINVAL;
  return jamd_gmm_create(e, &d, gprune, gprune_num, out);
}

int jamd_gms_load(jamd_engine *e, const char *path, jamd_gms **out) {
  if (!e || !path || !out) { jamd_set_error("jamd_gms_load: NULL argument"); return JAMD_EINVAL; }
  *out = nullptr;
  Blob b;
  if (!read_blob(path, "JAMDGMM1", b)) return JAMD_EINVAL;
  bool ok = true;
  const int *ints = view<int>(b, "ints", 0, 6, ok);
  const int *gms = view<int>(b, "gms", 0, 1, ok);
  if (!ok) { jamd_set_error("%s: not a selection model (no \"gms\" record)", path); return JAMD_EINVAL; }
  jamd_gmm_desc d;
  memset(&d, 0, sizeof(d));
  d.nstate = ints[0]; d.veclen = ints[1]; d.ndens = ints[2]; d.nentry = ints[3]; d.nbook = ints[4]; d.nstream = ints[5];
  if (d.nstate <= 0 || d.veclen <= 0 || d.ndens <= 0 || d.nentry <= 0) { jamd_set_error("%s: bad sizes", path); return JAMD_EINVAL; }
  d.mean = view<float>(b, "mean", 1, (long long)d.ndens * d.veclen, ok);
  d.ivar = view<float>(b, "ivar", 1, (long long)d.ndens * d.veclen, ok);
  d.gconst = view<float>(b, "gconst", 1, d.ndens, ok);
  d.st_off = view<int>(b, "st_off", 0, d.nstate + 1, ok);
  d.ent_dens = view<int>(b, "ent_dens", 0, d.nentry, ok);
  d.ent_logw = view<float>(b, "ent_logw", 1, d.nentry, ok);
  const int *map = view<int>(b, "state2gs", 0, -1, ok);
  if (!ok) return JAMD_EINVAL;
  return jamd_gms_create(e, &d, map, b["state2gs"].count, gms[0], out);
}

int jamd_rejgmm_load(jamd_engine *e, const char *path, jamd_rejgmm **out) {
  if (!e || !path || !out) { jamd_set_error("jamd_rejgmm_load: NULL argument"); return JAMD_EINVAL; }
  *out = nullptr;
  Blob b;
  if (!read_blob(path, "JAMDGMM1", b)) return JAMD_EINVAL;
  bool ok = true;
  const int *ints = view<int>(b, "ints", 0, 6, ok);
  const int *rej = view<int>(b, "rej", 0, 1, ok);
  if (!ok) { jamd_set_error("%s: not a verification-GMM file (no \"rej\" record)", path); return JAMD_EINVAL; }
  jamd_gmm_desc d;
  memset(&d, 0, sizeof(d));
  d.nstate = ints[0]; d.veclen = ints[1]; d.ndens = ints[2]; d.nentry = ints[3]; d.nbook = ints[4]; d.nstream = ints[5];
  if (d.nstate <= 0 || d.veclen <= 0 || d.ndens <= 0 || d.nentry <= 0) { jamd_set_error("%s: bad sizes", path); return JAMD_EINVAL; }
  d.mean = view<float>(b, "mean", 1, (long long)d.ndens * d.veclen, ok);
  d.ivar = view<float>(b, "ivar", 1, (long long)d.ndens * d.veclen, ok);
  d.gconst = view<float>(b, "gconst", 1, d.ndens, ok);
  d.st_off = view<int>(b, "st_off", 0, d.nstate + 1, ok);
  d.ent_dens = view<int>(b, "ent_dens", 0, d.nentry, ok);
  d.ent_logw = view<float>(b, "ent_logw", 1, d.nentry, ok);
  const int *ms = view<int>(b, "model_state", 0, -1, ok);
  if (!ok) return JAMD_EINVAL;
  const int nmodel = b["model_state"].count;
  const unsigned char *voice = view<unsigned char>(b, "is_voice", 2, nmodel, ok);
  const unsigned char *nm = view<unsigned char>(b, "model_names", 2, -1, ok);
  if (!ok) return JAMD_EINVAL;
  std::vector<std::string> names;
  {
    const int len = b["model_names"].count;
    int at = 0;
    while (at < len && (int)names.size() < nmodel) {
      const char *p = (const char *)nm + at;
      const size_t l = strnlen(p, (size_t)(len - at));
      names.emplace_back(p, l);
      at += (int)l + 1;
    }
    if ((int)names.size() != nmodel) { jamd_set_error("%s: %d model names for %d models", path, (int)names.size(), nmodel); return JAMD_EINVAL; }
  }
  int rc = jamd_rejgmm_create(e, &d, ms, nmodel, rej[0], out);
  if (rc != JAMD_OK) return rc;
  std::vector<const char *> np(nmodel);
  for (int k = 0; k < nmodel; k++) np[k] = names[k].c_str();
  rc = jamd_rejgmm_set_models(*out, np.data(), voice);
  if (rc != JAMD_OK) { jamd_rejgmm_destroy(*out); *out = nullptr; }
  return rc;
}

// The lexicon file's N-gram half against a binary N-gram: same vocabulary (names in id order), and -- `same_tables` --
// the same first-pass tables.  Returns false with the error set when the vocabularies differ.
static bool ngram_matches(const Blob &b, const char *path, const JamdNgramTables &t, const char *bingram, bool *same_tables) {
  auto nm = b.find("ng_wname");
  if (nm == b.end() || nm->second.dtype != 2) {
    jamd_set_error("%s carries no N-gram vocabulary (written by an older jamd_export): export it again", path); return false;
  }
  if ((size_t)nm->second.count != t.names.size() || memcmp(nm->second.data.data(), t.names.data(), t.names.size()) != 0) {
    jamd_set_error("%s was built for another vocabulary than %s (the tree's word -> N-gram ids and factoring values would not fit)", path, bingram);
    return false;
  }
  if (same_tables) {
    bool ok = true;
    auto same = [&](const char *name, const void *p, size_t bytes) {
      auto it = b.find(name);
      if (it == b.end() || (size_t)it->second.count * 4 != bytes || memcmp(it->second.data.data(), p, bytes) != 0) ok = false;
    };
    same("ng_uni_prob", t.uni_prob.data(), 4 * t.uni_prob.size()); same("ng_uni_bo", t.uni_bo.data(), 4 * t.uni_bo.size());
    same("ng_bi_bgn", t.bi_bgn.data(), 4 * t.bi_bgn.size()); same("ng_bi_num", t.bi_num.data(), 4 * t.bi_num.size());
    same("ng_bi_wid", t.bi_wid.data(), 4 * t.bi_wid.size()); same("ng_bi_prob", t.bi_prob.data(), 4 * t.bi_prob.size());
    *same_tables = ok;
  }
  return true;
}

// ---- a retrained N-gram under an exported tree ---------------------------------------------------------------------
// What wchmm.c derives from the 1-gram when it builds the tree (libjulius/src/wchmm.c:1749-1900, factoring_sub.c:345-468):
//  (1) WHICH words stay out of the tree: those whose p(w) = uni_prob(wton[w]) + cprob[w] reaches the separate_wnum-th best
//      value (get_nbest_uniprob(), wchmm.c:1470), taken in the builder's word order until separate_wnum are out
//      (:1882) -- for the same candidate set the builder makes the same tree;
//  (2) the 1-gram factoring value of every shared node: the best p(w) over the words that pass through it (:429-463).
// candidates() restates (1); refresh_fscore() restates (2) over the flattened tree -- a word passes through a node iff
// its word-end node lies in the node's subtree (self loops aside the arcs form a tree).
static float uni_of(const float *uni, int unk_id, float unk_num_log, int w) {      // ngram_access.c:229-236
  return w != unk_id ? uni[w] : uni[w] - unk_num_log;
}

static std::vector<unsigned char> candidates(const jamd_lexicon_desc &d, const float *uni, int sepnum) {
  const int W = d.nword;
  std::vector<float> p((size_t)W);
  for (int w = 0; w < W; w++) p[(size_t)w] = uni_of(uni, d.ng_unk_id, d.ng_unk_num_log, d.wton[w]) + d.cprob[w];
  std::vector<float> sorted(p);
  std::sort(sorted.begin(), sorted.end(), [](float a, float b) { return a > b; });
  int n = sepnum < 1 ? 1 : sepnum;
  if (n > W) n = W;
  const float thres = sorted[(size_t)n - 1];
  std::vector<unsigned char> c((size_t)W, 0);
  if (sepnum > 0)
    for (int w = 0; w < W; w++) c[(size_t)w] = (w != d.head_silwid && w != d.tail_silwid && p[(size_t)w] >= thres) ? 1 : 0;
  return c;
}

static bool refresh_fscore(const jamd_lexicon_desc &d, const float *uni, std::vector<float> &fs) {
  const int N = d.nnode;
  std::vector<float> best((size_t)N, JAMD_LOG_ZERO);
  std::vector<unsigned char> state((size_t)N, 0);             // 0 = new, 1 = on the stack, 2 = done
  std::vector<int> stack, child_at((size_t)N, 0);
  auto nchild = [&](int n) { return (d.next_a[n] != JAMD_LOG_ZERO && n + 1 < N ? 1 : 0) + (d.ac_off[n + 1] - d.ac_off[n]); };
  auto child = [&](int n, int k) {
    const int hasnext = (d.next_a[n] != JAMD_LOG_ZERO && n + 1 < N) ? 1 : 0;
    return (hasnext && k == 0) ? n + 1 : d.ac_to[d.ac_off[n] + k - hasnext];
  };
  fs.assign((size_t)(d.nfscore > 0 ? d.nfscore : 1), JAMD_LOG_ZERO);
  for (int root = 0; root < N; root++) {
    if (d.scid[root] >= 0 || state[(size_t)root] == 2) continue;
    stack.push_back(root); state[(size_t)root] = 1;
    while (!stack.empty()) {
      const int n = stack.back();
      if (child_at[(size_t)n] < nchild(n)) {
        const int c = child(n, child_at[(size_t)n]++);
        if (c == n || c < 0 || c >= N) continue;
        if (state[(size_t)c] == 1) return false;               // a cycle beyond self loops: not the tree this restatement assumes
        if (state[(size_t)c] == 0) { state[(size_t)c] = 1; stack.push_back(c); }
        continue;
      }
      float b = JAMD_LOG_ZERO;
      if (d.stend[n] >= 0 && d.stend[n] < d.nword)
        b = uni_of(uni, d.ng_unk_id, d.ng_unk_num_log, d.wton[d.stend[n]]) + d.cprob[d.stend[n]];
      for (int k = 0; k < nchild(n); k++) { const int c = child(n, k); if (c != n && c >= 0 && c < N && best[(size_t)c] > b) b = best[(size_t)c]; }
      best[(size_t)n] = b; state[(size_t)n] = 2; stack.pop_back();
    }
  }
  for (int n = 0; n < N; n++)
    if (d.scid[n] < 0) {
      if (-d.scid[n] >= d.nfscore) return false;
      fs[(size_t)(-d.scid[n])] = best[(size_t)n];
    }
  return true;
}

// PREFIX.lex -> descriptor (views into the blob `b`); with `bingram` the N-gram half comes from that binary N-gram (`ng`
// holds its tables, `fscore_new` the factoring values recomputed for a retrained 1-gram).
static int lexicon_desc_from_file(Blob &b, const char *path, const char *bingram, JamdNgramTables &ng, std::vector<float> &fscore_new,
                                  jamd_lexicon_desc &d) {
  if (!read_blob(path, "JAMDLEX1", b)) return JAMD_EINVAL;
  if (bingram) {
    if (!jamd_read_bingram_tables(bingram, ng)) return JAMD_EINVAL;
    if (!ngram_matches(b, path, ng, bingram, nullptr)) return JAMD_EINVAL;
  }
  bool ok = true;
  auto ir = b.find("ints"); auto fr = b.find("floats");
  if (ir == b.end() || fr == b.end() || ir->second.dtype != 0 || fr->second.dtype != 1 || ir->second.count < 18 || fr->second.count < 4) {
    jamd_set_error("%s: scalar records missing", path); return JAMD_EINVAL;
  }
  const int *I = reinterpret_cast<const int *>(ir->second.data.data());
  const float *F = reinterpret_cast<const float *>(fr->second.data.data());
  memset(&d, 0, sizeof(d));
  d.nnode = I[0]; d.nword = I[1]; d.startnum = I[2]; d.isolatenum = I[3]; d.nlc = I[4]; d.nlcrow = I[5]; d.nset = I[6];
  d.cdset_method = I[7]; d.cdmax_num = I[8]; d.head_silwid = I[9]; d.tail_silwid = I[10]; d.nfscore = I[11]; d.nscword = I[12];
  d.ng_mode = I[13]; d.ng_nword = I[14]; d.ng_nbigram = I[15]; d.ng_unk_id = I[16];
  if (ir->second.count >= 21) { d.lm_type = I[18]; d.ncat = I[19]; d.ninit = I[20]; }      // files written before grammar support stop at 18
  if (ir->second.count >= 22) d.nfwd = I[21];                                                // forward DFA (round 5)
  d.ng_unk_num_log = F[0]; d.lm_weight = F[1]; d.lm_penalty = F[2]; d.lm_penalty_trans = F[3];
  if (fr->second.count >= 5) d.penalty1 = F[4];
  if (d.nnode <= 0 || d.nword <= 0 || d.startnum < 0 || d.nset < 0 || d.nlc < 0 || d.nlcrow < 0) { jamd_set_error("%s: bad sizes", path); return JAMD_EINVAL; }
  d.self_a = view<float>(b, "self_a", 1, d.nnode, ok); d.next_a = view<float>(b, "next_a", 1, d.nnode, ok);
  d.ac_off = view<int>(b, "ac_off", 0, d.nnode + 1, ok);
  const long long nac = ok ? d.ac_off[d.nnode] : 0;
  d.ac_to = view<int>(b, "ac_to", 0, nac, ok); d.ac_a = view<float>(b, "ac_a", 1, nac, ok);
  d.stend = view<int>(b, "stend", 0, d.nnode, ok); d.scid = view<int>(b, "scid", 0, d.nnode, ok);
  d.out_kind = view<unsigned char>(b, "out_kind", 2, d.nnode, ok); d.out_id = view<int>(b, "out_id", 0, d.nnode, ok);
  d.lc_tab = view<int>(b, "lc_tab", 0, (long long)d.nlcrow * (d.nlc + 1), ok); d.word_lc = view<int>(b, "word_lc", 0, d.nword, ok);
  d.set_off = view<int>(b, "set_off", 0, d.nset + 1, ok);
  d.set_states = view<int>(b, "set_states", 0, ok ? d.set_off[d.nset] : 0, ok);
  d.startnode = view<int>(b, "startnode", 0, d.startnum, ok); d.start2isolate = view<int>(b, "start2isolate", 0, d.startnum, ok);
  d.wordend_a = view<float>(b, "wordend_a", 1, d.nword, ok); d.wton = view<int>(b, "wton", 0, d.nword, ok);
  d.cprob = view<float>(b, "cprob", 1, d.nword, ok);
  d.is_transparent = view<unsigned char>(b, "is_transparent", 2, d.nword, ok); d.word_head = view<int>(b, "word_head", 0, d.nword, ok);
  d.fscore = view<float>(b, "fscore", 1, d.nfscore, ok); d.scword = view<int>(b, "scword", 0, d.nscword, ok);
  d.ng_uni_prob = view<float>(b, "ng_uni_prob", 1, d.ng_nword, ok); d.ng_uni_bo = view<float>(b, "ng_uni_bo", 1, d.ng_nword, ok);
  d.ng_bi_bgn = view<int>(b, "ng_bi_bgn", 0, d.ng_nword, ok); d.ng_bi_num = view<int>(b, "ng_bi_num", 0, d.ng_nword, ok);
  d.ng_bi_wid = view<int>(b, "ng_bi_wid", 0, d.ng_nbigram, ok); d.ng_bi_prob = view<float>(b, "ng_bi_prob", 1, d.ng_nbigram, ok);
  if ((d.lm_type & 0xff) != JAMD_LM_NGRAM) {                   // (JAMD_LM_MULTIPATH is a flag beside the LM kind)
    d.cat_pair = view<unsigned char>(b, "cat_pair", 2, (long long)d.ncat * d.ncat, ok);
    d.start2wid = view<int>(b, "start2wid", 0, d.startnum, ok);
    d.init_node = view<int>(b, "init_node", 0, d.ninit, ok); d.init_lscore = view<float>(b, "init_lscore", 1, d.ninit, ok);
    if (d.nfwd > 0) {
      d.fwd_off = view<int>(b, "fwd_off", 0, (long long)d.nfwd + 1, ok);
      const long long nfa = ok ? d.fwd_off[d.nfwd] : 0;
      d.fwd_label = view<int>(b, "fwd_label", 0, nfa, ok); d.fwd_to = view<int>(b, "fwd_to", 0, nfa, ok);
      d.init_to_state = view<int>(b, "init_to_state", 0, d.ninit, ok);
    }
  } else d.nfwd = 0;
  if (!ok) return JAMD_EINVAL;
  if (bingram) {
    // the N-gram half from the binary N-gram itself (libsent/src/ngram/ngram_read_bin.c:240-365): 1-gram and 2-gram
    // tables, which 2-gram the first pass reads; the cross-word LM table is built from them when the lexicon is created.
    // The tree half -- nodes, word -> N-gram ids, class probabilities, factoring values -- stays the file's.
    if ((d.lm_type & 0xff) != JAMD_LM_NGRAM) { jamd_set_error("%s is a grammar lexicon: no N-gram to replace", path); return JAMD_EINVAL; }
    if (ng.nword != d.ng_nword) { jamd_set_error("%s: %d N-gram words, %s has %d", path, d.ng_nword, bingram, ng.nword); return JAMD_EINVAL; }
    if (memcmp(ng.uni_prob.data(), d.ng_uni_prob, sizeof(float) * (size_t)d.ng_nword) != 0) {
      // a RETRAINED 1-gram: the tree half depends on it in two places (see candidates() / refresh_fscore() above)
      auto sp = b.find("sep_wnum");
      if (sp == b.end() || sp->second.dtype != 0 || sp->second.count < 1) {
        jamd_set_error("%s does not record -sepnum (written by an older jamd_export): a retrained N-gram needs it, export the lexicon again", path);
        return JAMD_EINVAL;
      }
      const int sepnum = *reinterpret_cast<const int *>(sp->second.data.data());
      if (candidates(d, d.ng_uni_prob, sepnum) != candidates(d, ng.uni_prob.data(), sepnum)) {
        jamd_set_error("%s: under the 1-gram of %s other words are the %d most frequent ones, which wchmm.c keeps out of the tree "
                       "(-sepnum): the tree itself would differ, export the lexicon again", path, bingram, sepnum);
        return JAMD_EINVAL;
      }
      std::vector<float> check;
      if (!refresh_fscore(d, d.ng_uni_prob, check) || (size_t)d.nfscore > check.size() ||
          memcmp(check.data() + 1, d.fscore + 1, sizeof(float) * (size_t)(d.nfscore > 1 ? d.nfscore - 1 : 0)) != 0) {
        jamd_set_error("%s: the factoring values of this tree are not the subtree maxima of its own 1-gram (a lexicon form the "
                       "refresh does not cover): export the lexicon again with the new N-gram", path);
        return JAMD_EINVAL;
      }
      if (!refresh_fscore(d, ng.uni_prob.data(), fscore_new)) { jamd_set_error("%s: factoring refresh failed", path); return JAMD_EINVAL; }
      d.fscore = fscore_new.data();
    }
    d.ng_mode = ng.mode; d.ng_nword = ng.nword; d.ng_nbigram = ng.nbigram;
    d.ng_uni_prob = ng.uni_prob.data(); d.ng_uni_bo = ng.uni_bo.data();
    d.ng_bi_bgn = ng.bi_bgn.data(); d.ng_bi_num = ng.bi_num.data(); d.ng_bi_wid = ng.bi_wid.data(); d.ng_bi_prob = ng.bi_prob.data();
  }
  return JAMD_OK;
}

static int load_lexicon(jamd_engine *e, const char *path, const char *bingram, jamd_lexicon **out) {
  *out = nullptr;
  Blob b;
  JamdNgramTables ng;
  std::vector<float> fscore_new;
  jamd_lexicon_desc d;
  const int rc = lexicon_desc_from_file(b, path, bingram, ng, fscore_new, d);
  if (rc != JAMD_OK) return rc;
  return jamd_lexicon_create(e, &d, out);
}

int jamd_lexicon_load(jamd_engine *e, const char *path, jamd_lexicon **out) {
  if (!e || !path || !out) { jamd_set_error("jamd_lexicon_load: NULL argument"); return JAMD_EINVAL; }
  return load_lexicon(e, path, nullptr, out);
}

int jamd_lexicon_load_ngram(jamd_engine *e, const char *path, const char *bingram_path, jamd_lexicon **out) {
  if (!e || !path || !bingram_path || !out) { jamd_set_error("jamd_lexicon_load_ngram: NULL argument"); return JAMD_EINVAL; }
  return load_lexicon(e, path, bingram_path, out);
}

int jamd_bingram_fscore(const char *lex_path, const char *bingram_path, float *fscore, int cap, int *nfscore) {
  if (!lex_path || !bingram_path || !nfscore || (cap > 0 && !fscore)) { jamd_set_error("jamd_bingram_fscore: NULL argument"); return JAMD_EINVAL; }
  Blob b;
  JamdNgramTables ng;
  std::vector<float> fscore_new;
  jamd_lexicon_desc d;
  const int rc = lexicon_desc_from_file(b, lex_path, bingram_path, ng, fscore_new, d);
  if (rc != JAMD_OK) return rc;
  *nfscore = d.nfscore;
  for (int i = 0; i < d.nfscore && i < cap; i++) fscore[i] = d.fscore[i];
  return JAMD_OK;
}

int jamd_bingram_check(const char *lex_path, const char *bingram_path, int *same_tables) {
  if (!lex_path || !bingram_path) { jamd_set_error("jamd_bingram_check: NULL argument"); return JAMD_EINVAL; }
  Blob b;
  if (!read_blob(lex_path, "JAMDLEX1", b)) return JAMD_EINVAL;
  JamdNgramTables ng;
  if (!jamd_read_bingram_tables(bingram_path, ng)) return JAMD_EINVAL;
  bool same = false;
  if (!ngram_matches(b, lex_path, ng, bingram_path, &same)) return JAMD_EINVAL;
  if (same_tables) *same_tables = same ? 1 : 0;
  return JAMD_OK;
}

int jamd_dnn_load(jamd_engine *e, const char *dnnconf, jamd_dnn **out) {
  if (!e || !dnnconf || !out) { jamd_set_error("jamd_dnn_load: NULL argument"); return JAMD_EINVAL; }
  *out = nullptr;
  FILE *f = fopen(dnnconf, "r");
  if (!f) { jamd_set_error("cannot open %s", dnnconf); return JAMD_EINVAL; }
  const std::string dir = dir_of(dnnconf);
  int veclen = 0, contextlen = 0, in = 0, outn = 0, hid = 0, nh = 0, log10nize = 1;
  float prior_factor = 1.0f;                              // jconf default (libjulius/src/default.c)
  std::vector<std::string> wf, bf;
  std::string ow, ob, prior;
  char line[4096];
  bool ok = true;
  while (ok && fgets(line, sizeof(line), f)) {            // dnn_config_file_parse(), m_jconf.c:601-690
    if (char *h = strchr(line, '#')) *h = 0;
    size_t n = strlen(line);
    while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r' || line[n - 1] == ' ' || line[n - 1] == '\t')) line[--n] = 0;
    char *k = line;
    while (*k == ' ' || *k == '\t') k++;
    if (*k == 0) continue;
    char *sp = strchr(k, ' ');
    if (!sp) { jamd_set_error("%s: wrong line: %s", dnnconf, k); ok = false; break; }
    char *v = sp;
    while (*v == ' ') v++;
    *sp = 0;
    const std::string key(k);
    if (key == "feature_len") veclen = atoi(v);
    else if (key == "context_len") contextlen = atoi(v);
    else if (key == "input_nodes") in = atoi(v);
    else if (key == "output_nodes") outn = atoi(v);
    else if (key == "hidden_nodes") hid = atoi(v);
    else if (key == "hidden_layers") { nh = atoi(v); if (nh < 0 || nh > 64) { ok = false; break; } wf.assign(nh, ""); bf.assign(nh, ""); }
    else if (key == "output_W") ow = rel(dir, v);
    else if (key == "output_B") ob = rel(dir, v);
    else if (key == "state_prior") prior = rel(dir, v);
    else if (key == "state_prior_factor") prior_factor = (float)atof(v);
    else if (key == "state_prior_log10nize") {
      if (!strcmp(v, "yes") || !strcmp(v, "true")) log10nize = 1;
      else if (!strcmp(v, "no") || !strcmp(v, "false")) log10nize = 0;
      else { jamd_set_error("%s: state_prior_log10nize must be true or false", dnnconf); ok = false; }
    } else if (key[0] == 'W' || key[0] == 'B') {
      const int l = atoi(k + 1);
      if (l <= 0 || l > nh) { jamd_set_error("%s: layer id %d outside 1..%d", dnnconf, l, nh); ok = false; }
      else (key[0] == 'W' ? wf : bf)[l - 1] = rel(dir, v);
    } else if (key == "feature_type" || key == "feature_options" || key == "batch_size" || key == "num_threads" || key == "cuda_mode") {
      // front-end / host threading / CUDA settings: not part of the model
    } else { jamd_set_error("%s: unknown spec: %s", dnnconf, k); ok = false; }
  }
  fclose(f);
  if (ok && (nh <= 0 || in <= 0 || outn <= 0 || hid <= 0 || ow.empty() || ob.empty() || prior.empty())) {
    jamd_set_error("%s: incomplete DNN definition", dnnconf); ok = false;
  }
  if (ok && veclen > 0 && contextlen > 0 && veclen * contextlen != in) {     // calc_dnn.c:640-643
    jamd_set_error("%s: veclen(%d) * contextlen(%d) != inputnodes(%d)", dnnconf, veclen, contextlen, in); ok = false;
  }
  for (int l = 0; ok && l < nh; l++) if (wf[l].empty() || bf[l].empty()) { jamd_set_error("%s: no W/B file for hidden layer #%d", dnnconf, l + 1); ok = false; }
  if (!ok) return JAMD_EINVAL;
  const int nlayer = nh + 1;
  std::vector<int> dims(nlayer + 1);
  dims[0] = in;
  for (int l = 1; l <= nh; l++) dims[l] = hid;
  dims[nlayer] = outn;
  std::vector<std::vector<float>> W(nlayer), B(nlayer);
  for (int l = 0; ok && l < nlayer; l++) {                // dnn_layer_load(), calc_dnn.c:390-412
    ok = read_npy(l < nh ? wf[l] : ow, (long long)dims[l] * dims[l + 1], W[l]) &&
         read_npy(l < nh ? bf[l] : ob, dims[l + 1], B[l]);
  }
  std::vector<float> pr(outn, 0.0f);
  if (ok) {                                               // calc_dnn.c:678-707
    FILE *pf = fopen(prior.c_str(), "r");
    if (!pf) { jamd_set_error("cannot open %s", prior.c_str()); ok = false; }
    else {
      int id; float val;
      while (ok && fscanf(pf, "%d %e", &id, &val) == 2) {
        if (id < 0 || id >= outn) { jamd_set_error("%s: wrong state id %d", prior.c_str(), id); ok = false; break; }
        pr[id] = val * prior_factor;
        if (log10nize) pr[id] = (float)log10(pr[id]);
      }
      fclose(pf);
    }
  }
  if (!ok) return JAMD_EINVAL;
  std::vector<const float *> wp(nlayer), bp(nlayer);
  for (int l = 0; l < nlayer; l++) { wp[l] = W[l].data(); bp[l] = B[l].data(); }
  jamd_dnn_desc d;
  d.nlayer = nlayer; d.dims = dims.data(); d.w = wp.data(); d.b = bp.data(); d.state_prior = pr.data();
  return jamd_dnn_create(e, &d, out);
}

}  // extern "C"
