/*
 * jamd_batch -- batch-of-utterances first pass over the C ABI, no Julius process.
 *
 *   jamd_batch [-d device | -devices 0,1,..|0-7] [-b beam] [-bs score_width] [-gprune none|safe N|heu N|beam N]
 *              [-order exact|fast|strict] [-shard R N] [-rej verification.blob] [-launch N] [-time]
 *              (-am model.blob [-gms selection.blob] | -dnnconf dnn.conf) -lex lexicon.blob -filelist list.txt
 *
 * -shard R N: this process takes the utterances u with u % N == R (one process per GPU, e.g.
 * `for r in 0..7: jamd_batch -d $r -shard $r 8 ...`): utterances are independent, the model is
 * replicated, nothing is exchanged.
 * -devices LIST: the same sharding INSIDE one process -- one host thread with its own engine, models and first-pass
 * work area per listed device (a device may be listed twice), utterance u of the file list on list entry u % n; the
 * result lines of all devices are merged and printed in file-list order when every device is done (BASELINE
 * configs[4]: one node, eight GPUs, one batch).  Combines with -shard (R N processes x n devices each).
 *
 * model.blob / lexicon.blob are written once by a Julius process through the shim
 * (jamd_gmm_save / jamd_lexicon_save); a DNN is read from Julius' own dnnconf + .npy files.
 * Every line of the file list names an HTK parameter file (what Julius reads with
 * `-input htkparam`: 12-byte big-endian header nSamples, sampPeriod, sampSize, parmKind, then
 * big-endian float vectors -- libsent/src/anlz/rdparam.c).  All utterances are scored and
 * decoded in device launches of up to 512 utterances, three launches in flight: while the first pass of launch k runs,
 * launch k+1 is scored (in the CUs the shorter utterances of launch k leave) and the files of launch k+2 are read into
 * pinned staging memory and uploaded; -time reports the process's own clock.  One result line per utterance:
 *   <file> status=<0 ok|1 no result|2 beam died|3 trellis overflow> score=<pass-1 score> words=<id id ...>
 * which is what get_back_trellis_end() leaves in r->pass1_wseq / pass1_score.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/stat.h>
#include "julius_amd.h"

static void die(const char *what)
{
  fprintf(stderr, "jamd_batch: %s: %s\n", what, jamd_last_error());
  exit(1);
}

static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* frames of an HTK parameter file of `veclen`-dim float vectors, from its SIZE (one stat() instead of open + read + close:
 * on a box whose file system makes an open cost 0.2 ms, two opens per 220 KB file were most of a launch's host time); the
 * header is checked when the file is read.  -1: not there / not a whole number of vectors. */
static int htk_frames(const char *path, int veclen)
{
  /* nSamples and sampSize come from the 12-byte HTK header (nSamples, sampPeriod: 4 bytes each; sampSize, parmKind: 2 each,
   * big endian); the body must hold at least nSamples vectors -- trailing bytes (a CRC word of _K files) are accepted, as
   * by the reference's reader (the size alone used to decide, and such files were refused). */
  unsigned char h[12];
  struct stat st;
  long n, size;
  int fd = open(path, O_RDONLY);
  ssize_t got;
  if (fd < 0) return -1;
  got = read(fd, h, 12);
  if (got != 12 || fstat(fd, &st) != 0) { close(fd); return -1; }
  close(fd);
  n = (long)be32(h);
  size = ((long)h[8] << 8) | (long)h[9];
  if (size != 4 * (long)veclen || n < 1 || n > 0x7fffffffL) return -1;
  if ((long)st.st_size - 12 < n * size) return -1;
  return (int)n;
}

/* One reader: files [u0, u1) of a launch -> their slots of the pinned staging buffer.  A file is read whole (header and
 * vectors, one open) into an ordinary cached bounce buffer, its header checked against what its size promised, and the
 * byte swap WRITES the pinned buffer once, front to back: pinned host memory may be mapped uncached for the CPU, and an
 * in-place swap would read it back at a few hundred MB/s. */
typedef struct {
  char **files; const int *off; float *frames; int veclen, u0, u1, rc;
  unsigned char *bounce; size_t bounce_cap;
} reader;

static void *reader_main(void *arg)
{
  reader *r = (reader *)arg;
  int u;
  r->rc = 0;
  for (u = r->u0; u < r->u1 && r->rc == 0; u++) {
    const int n = r->off[u + 1] - r->off[u];
    const size_t nfloat = (size_t)n * (size_t)r->veclen, bytes = 12 + 4 * nfloat;
    uint32_t *w = (uint32_t *)(r->frames + (size_t)r->off[u] * r->veclen);
    size_t got = 0, i;
    int fd;
    if (bytes > r->bounce_cap) {
      free(r->bounce);
      r->bounce_cap = bytes + bytes / 4 + 4096;
      r->bounce = (unsigned char *)malloc(r->bounce_cap);
      if (r->bounce == NULL) { r->bounce_cap = 0; r->rc = -1 - u; break; }
    }
    if ((fd = open(r->files[u], O_RDONLY)) < 0) { r->rc = -1 - u; break; }
    while (got < bytes) { const ssize_t k = read(fd, r->bounce + got, bytes - got); if (k <= 0) break; got += (size_t)k; }
    close(fd);
    if (got != bytes || (int)be32(r->bounce) != n || (((int)r->bounce[8] << 8) | r->bounce[9]) != 4 * r->veclen) { r->rc = -1 - u; break; }
    for (i = 0; i < nfloat; i++) { uint32_t v; memcpy(&v, r->bounce + 12 + 4 * i, 4); w[i] = __builtin_bswap32(v); }
  }
  return NULL;
}

/* One launch: up to LAUNCH utterances, their frames on the host and on the device, their score rows.  512 = two per CU
 * of an MI355X: the exact-order first pass then runs in its half shape (jamd_beam_set_workgroup_shape(), automatic),
 * the one with the most frames per second; beams too wide for it run the same launch one utterance per CU.
 * The two chunks of a device are its double buffer: PINNED host staging (jamd_host_alloc(): the upload is a true
 * asynchronous DMA, the host goes on reading the next files) and the device buffers, all kept from launch to launch and
 * grown only when a launch needs more (no allocation -- which would synchronise the device -- in the steady state). */
#define LAUNCH 512
typedef struct {
  float *frames, *d_frames, *d_scores;      /* pinned staging, device frames, device score rows */
  size_t cap_frames, cap_scores;            /* their capacities in bytes */
  int off[LAUNCH + 1], n;
} chunk;

static int launch = LAUNCH;      /* -launch N: utterances per device launch (1 .. LAUNCH) */
static int want_time = 0;        /* -time: one JSON line per device on stderr (model load, decode wall time, host read time) */

#define READERS 4                 /* host threads that read and byte-swap the files of a launch side by side */
typedef struct { double read_s, sync_s; size_t h2d_bytes; int launches; reader rd[READERS]; } hosttime;

/* reads the files of the launch that starts at files[first] into the chunk's pinned buffer and queues the upload of
 * the frames on `stream` (nothing is waited for) */
/* the chunk's buffers for a launch of need_fr bytes of frames and need_sc bytes of scores (pinning 100+ MB and a multi-GB
 * hipMalloc take 0.1 - 0.3 s: run_device() does it for the largest launch of the list before the decode clock starts) */
static void reserve(jamd_engine *e, chunk *c, size_t need_fr, size_t need_sc)
{
  if (need_fr > c->cap_frames) {
    if (c->frames) jamd_host_free(e, c->frames);
    if (c->d_frames) jamd_free(e, c->d_frames);
    c->cap_frames = need_fr + need_fr / 16;
    if (jamd_host_alloc(e, c->cap_frames, (void **)&c->frames) != JAMD_OK ||
        jamd_malloc(e, c->cap_frames, (void **)&c->d_frames) != JAMD_OK) die("frame buffers");
  }
  if (need_sc > c->cap_scores) {
    if (c->d_scores) jamd_free(e, c->d_scores);
    c->cap_scores = need_sc + need_sc / 16;
    if (jamd_malloc(e, c->cap_scores, (void **)&c->d_scores) != JAMD_OK) die("score buffer");
  }
}

static void load(jamd_engine *e, chunk *c, char **files, const int *nfr, int first, int nfile, int veclen, int nstate, void *stream, hosttime *ht)
{
  const double t0 = now_s();
  size_t need_fr, need_sc;
  int u;
  c->n = nfile - first < launch ? nfile - first : launch;
  c->off[0] = 0;
  for (u = 0; u < c->n; u++) {
    const int t = nfr[first + u] >= 0 ? nfr[first + u] : htk_frames(files[first + u], veclen);   /* (counted once, in the pre-sizing pass) */
    if (t < 0) { fprintf(stderr, "jamd_batch: cannot read %s as %d-dim HTK parameters\n", files[first + u], veclen); exit(1); }
    c->off[u + 1] = c->off[u] + t;
  }
  need_fr = sizeof(float) * (size_t)c->off[c->n] * (size_t)veclen;
  need_sc = sizeof(float) * (size_t)c->off[c->n] * (size_t)nstate;
  reserve(e, c, need_fr, need_sc);                     /* (sized before the clock starts, run_device(): a no-op here unless files changed) */
  {
    pthread_t th[READERS];
    int r, started = 0;
    for (r = 0; r < READERS; r++) {
      reader *rd = &ht->rd[r];
      rd->files = files + first; rd->off = c->off; rd->frames = c->frames; rd->veclen = veclen;
      rd->u0 = (int)((long)c->n * r / READERS); rd->u1 = (int)((long)c->n * (r + 1) / READERS);
      if (r + 1 < READERS && pthread_create(&th[r], NULL, reader_main, rd) == 0) started |= 1 << r;
      else reader_main(rd);                            /* the last share (and any share whose thread did not start) on this thread */
    }
    for (r = 0; r < READERS; r++) if (started & (1 << r)) pthread_join(th[r], NULL);
    for (r = 0; r < READERS; r++)
      if (ht->rd[r].rc != 0) { fprintf(stderr, "jamd_batch: cannot read %s as %d-dim HTK parameters\n", files[first - 1 - ht->rd[r].rc], veclen); exit(1); }
  }
  if (jamd_memcpy_h2d_async(e, c->d_frames, c->frames, need_fr, stream) != JAMD_OK) die("upload");
  ht->read_s += now_s() - t0; ht->h2d_bytes += need_fr; ht->launches++;
}

/* queues the scoring kernels of a loaded launch on `stream` */
static void score(chunk *c, int nstate, jamd_gmm *gm, jamd_dnn *dn, jamd_gms *gs, void *stream)
{
  (void)nstate;
  if ((gm ? jamd_gmm_outprob_utts_dev(gm, c->d_frames, c->off, c->n, c->d_scores, stream)
          : jamd_dnn_outprob_dev(dn, c->d_frames, c->off[c->n], c->d_scores, stream)) != JAMD_OK) die("scoring");
  if (gs != NULL && jamd_gms_apply_dev(gs, c->d_frames, c->off[c->n], c->off, c->n, c->d_scores, stream) != JAMD_OK) die("Gaussian mixture selection");
}

/* options (shared by the device threads, read-only once parsed) */
static const char *am = NULL, *dnnconf = NULL, *lexp = NULL, *list = NULL, *gmsp = NULL, *rejp = NULL;
static int beam = 800, gprune = JAMD_GPRUNE_NONE, gnum = 0, strict = 0, order = -1;
static float bs = -1.0f;

/* one device's share of the file list: files[0..nfile) are ITS utterances, gidx[] their numbers in the whole list; with
 * out != NULL the result lines go to out[gidx[u]] (merged by main), else they are printed as the launches complete */
typedef struct { int device; char **files; int *gidx; int nfile; char **out; } devjob;

static void emit(devjob *j, int u, const char *text)
{
  if (j->out != NULL) j->out[j->gidx[u]] = strdup(text);
  else fputs(text, stdout);
}

static void *run_device(void *arg)
{
  devjob *j = (devjob *)arg;
  char **files = j->files;
  const int nfile = j->nfile;
  jamd_engine *e; jamd_gmm *gm = NULL; jamd_dnn *dn = NULL; jamd_gms *gs = NULL; jamd_rejgmm *rj = NULL; jamd_lexicon *lx; jamd_beam *bm;
  int veclen, nstate, first, k;
  chunk ck[3];
  hosttime ht;
  double t_start = now_s(), t_models, t_done;
  long frames_total = 0;
  void *s_copy = NULL, *s_beam = NULL;
  size_t linecap = 1 << 16;
  char *text = (char *)malloc(linecap);
  int *nfr = (int *)malloc(sizeof(int) * (size_t)(nfile > 0 ? nfile : 1));   /* frames per file, counted once */

  if (text == NULL || nfr == NULL) die("out of memory");
  for (k = 0; k < nfile; k++) nfr[k] = -1;
  memset(&ht, 0, sizeof(ht));
  if (jamd_engine_create(j->device, &e) != JAMD_OK) die("engine");
  if (am != NULL) { if (jamd_gmm_load(e, am, gprune, gnum, &gm) != JAMD_OK) die("acoustic model"); }
  else if (jamd_dnn_load(e, dnnconf, &dn) != JAMD_OK) die("DNN");
  veclen = gm ? jamd_gmm_veclen(gm) : jamd_dnn_veclen(dn);
  nstate = gm ? jamd_gmm_nstate(gm) : jamd_dnn_nstate(dn);
  if (gmsp != NULL) {                                 /* -gshmm of the exported configuration */
    if (jamd_gms_load(e, gmsp, &gs) != JAMD_OK) die("selection model");
    if (strict && jamd_gms_set_strict_order(gs, 1) != JAMD_OK) die("strict order");
    if (jamd_gms_nstate(gs) != nstate) { fprintf(stderr, "jamd_batch: %s belongs to another acoustic model\n", gmsp); exit(1); }
  }
  if (rejp != NULL) {                                 /* -gmm / -gmmnum / -gmmreject of the exported configuration */
    if (jamd_rejgmm_load(e, rejp, &rj) != JAMD_OK) die("verification GMMs");
    if (jamd_rejgmm_veclen(rj) != veclen) { fprintf(stderr, "jamd_batch: %s is for %d-dim input\n", rejp, jamd_rejgmm_veclen(rj)); exit(1); }
  }
  if (jamd_lexicon_load(e, lexp, &lx) != JAMD_OK) die("lexicon");
  if (jamd_beam_create(e, lx, beam, bs, LAUNCH, 1 << 18, &bm) != JAMD_OK) die("first-pass work area");
  if (strict && jamd_beam_set_strict_order(bm, 1) != JAMD_OK) die("strict order");
  if (!strict && order >= 0 && jamd_beam_set_order_mode(bm, order) != JAMD_OK) die("order mode");

  /* Launches of up to LAUNCH utterances over two streams: `s_copy` uploads and scores, `s_beam` searches; THREE chunks in
   * flight: while the first pass of launch k runs, launch k+1 (read and uploaded one iteration earlier) is scored -- queued
   * at once behind jamd_beam_stream_wait_resident(): the scoring starts when that first pass holds its CUs (queued any
   * earlier, the scoring workgroups take the LDS the first pass wants for its second utterance per CU and it takes twice
   * as long) and fills the CUs that the shorter utterances of launch k leave -- and the host reads the files of launch
   * k+2.  File reading is then off the device's critical path whatever the file system does (round 5: with two chunks a
   * box with slow reads delayed the QUEUEING of the next scoring by the read time: 442 instead of 300 ms per launch). */
  if (jamd_stream_create(e, &s_copy) != JAMD_OK || jamd_stream_create(e, &s_beam) != JAMD_OK) die("streams");
  memset(ck, 0, sizeof(ck));
  {
    /* the staging and device buffers of the (at most three) launches in flight, sized for the largest launch of the list
     * (one stat() per file): set-up, like the models -- the steady state allocates nothing */
    size_t most = 0;
    int nl = 0, f0;
    for (f0 = 0; f0 < nfile; f0 += launch, nl++) {
      size_t fr = 0;
      int u2;
      for (u2 = f0; u2 < nfile && u2 < f0 + launch; u2++) {
        const int t = nfr[u2] = htk_frames(files[u2], veclen);
        if (t < 0) { fprintf(stderr, "jamd_batch: cannot read %s as %d-dim HTK parameters\n", files[u2], veclen); exit(1); }
        fr += (size_t)t;
      }
      if (fr > most) most = fr;
    }
    for (f0 = 0; f0 < 3 && f0 < nl; f0++) reserve(e, &ck[f0], sizeof(float) * most * (size_t)veclen, sizeof(float) * most * (size_t)nstate);
  }
  if (jamd_engine_sync(e) != JAMD_OK) die("model upload");
  t_models = now_s();                                  /* engine + models + work area are up: the decode clock starts */
  if (nfile > 0) {
    load(e, &ck[0], files, nfr, 0, nfile, veclen, nstate, s_copy, &ht); score(&ck[0], nstate, gm, dn, gs, s_copy);
    if (jamd_stream_wait(e, s_beam, s_copy) != JAMD_OK) die("stream order");                   /* the scores of launch 0 */
  }
  for (first = 0, k = 0; first < nfile; first += launch, k++) {
    chunk *c = &ck[k % 3];
    const int n = c->n;
    const int *off = c->off;
    int u;
    float *us = NULL;
    jamd_pass1_result *res = (jamd_pass1_result *)malloc(sizeof(jamd_pass1_result) * LAUNCH);
    if (res == NULL) die("out of memory");
    if (jamd_beam_pass1_dev(bm, c->d_scores, nstate, off, n, s_beam) != JAMD_OK) die("first pass");
    if (first + launch < nfile) {                      /* launch k+1: its frames are on their way or there already */
      chunk *nx = &ck[(k + 1) % 3];
      if (k == 0) load(e, nx, files, nfr, launch, nfile, veclen, nstate, s_copy, &ht);   /* (the first pass of launch 0 is queued: now read launch 1) */
      if (jamd_beam_stream_wait_resident(bm, s_copy) != JAMD_OK) die("first pass");   /* s_copy goes on once the first pass holds its CUs */
      score(nx, nstate, gm, dn, gs, s_copy);
      if (jamd_stream_wait(e, s_beam, s_copy) != JAMD_OK) die("stream order");        /* the next first pass waits for exactly these scores */
    }
    if (first + 2 * launch < nfile)                    /* launch k+2: read and upload while the device is busy with k and k+1 */
      load(e, &ck[(k + 2) % 3], files, nfr, first + 2 * launch, nfile, veclen, nstate, s_copy, &ht);
    { const double tw = now_s();
      if (jamd_stream_sync(e, s_beam) != JAMD_OK || jamd_beam_results(bm, res, n) != JAMD_OK) die("first pass");
      ht.sync_s += now_s() - tw; frames_total += off[n]; }
    if (rj != NULL) {                                  /* gmm_proceed() over every frame, gmm_end() per input */
      const int nm = jamd_rejgmm_nmodel(rj);
      float *d_fs = NULL, *d_us = NULL;
      us = (float *)malloc(sizeof(float) * (size_t)n * nm);
      if (us == NULL || jamd_malloc(e, sizeof(float) * (size_t)off[n] * nm, (void **)&d_fs) != JAMD_OK ||
          jamd_malloc(e, sizeof(float) * (size_t)n * nm, (void **)&d_us) != JAMD_OK ||
          jamd_rejgmm_frame_scores_dev(rj, c->d_frames, off[n], d_fs, s_beam) != JAMD_OK ||
          jamd_rejgmm_utt_scores_dev(rj, d_fs, off[n], off, n, d_us, s_beam) != JAMD_OK ||
          jamd_stream_sync(e, s_beam) != JAMD_OK || jamd_memcpy_d2h(e, us, d_us, sizeof(float) * (size_t)n * nm) != JAMD_OK) die("input verification");
      jamd_free(e, d_fs); jamd_free(e, d_us);
    }
    for (u = 0; u < n; u++) {
      int w;
      size_t at = 0;
      if ((size_t)res[u].wnum * 12 + strlen(files[first + u]) + 512 > linecap) { linecap = 2 * ((size_t)res[u].wnum * 12 + strlen(files[first + u]) + 512); text = (char *)realloc(text, linecap); if (text == NULL) die("out of memory"); }
      at += (size_t)sprintf(text + at, "%s status=%d score=%.9g words=", files[first + u], res[u].status, (double)res[u].score);
      for (w = 0; w < res[u].wnum; w++) at += (size_t)sprintf(text + at, "%s%d", w ? " " : "", res[u].wseq[w]);
      if (rj != NULL) {
        int win, acc; float cm;
        if (jamd_rejgmm_verdict(rj, us + (size_t)u * jamd_rejgmm_nmodel(rj), &win, &cm, &acc) != JAMD_OK) die("verdict");
        at += (size_t)sprintf(text + at, " gmm=%s gmmscore=%.9g cm=%.9g accepted=%d", jamd_rejgmm_model_name(rj, win),
                              (double)us[(size_t)u * jamd_rejgmm_nmodel(rj) + win], (double)cm, acc);
      }
      sprintf(text + at, "\n");
      emit(j, first + u, text);
    }
    free(us); free(res);
  }
  t_done = now_s();
  if (want_time)       /* decode_s: first file read -> last result line, everything in between (reads, uploads, scoring, first pass, result copies) */
    fprintf(stderr, "{\"jamd_batch_time\": {\"device\": %d, \"utts\": %d, \"frames\": %ld, \"launches\": %d, \"models_s\": %.6f, "
                    "\"decode_s\": %.6f, \"host_read_s\": %.6f, \"host_wait_s\": %.6f, \"h2d_bytes\": %zu, \"rtf_inv\": %.3f}}\n",
            j->device, nfile, frames_total, ht.launches, t_models - t_start, t_done - t_models, ht.read_s, ht.sync_s, ht.h2d_bytes,
            t_done > t_models ? (double)frames_total / 100.0 / (t_done - t_models) : 0.0);
  for (k = 0; k < 3; k++) {
    if (ck[k].frames) jamd_host_free(e, ck[k].frames);
    if (ck[k].d_frames) jamd_free(e, ck[k].d_frames);
    if (ck[k].d_scores) jamd_free(e, ck[k].d_scores);
  }
  { int r; for (r = 0; r < READERS; r++) free(ht.rd[r].bounce); }
  jamd_stream_destroy(e, s_copy); jamd_stream_destroy(e, s_beam);
  jamd_beam_destroy(bm); jamd_lexicon_destroy(lx);
  if (rj) jamd_rejgmm_destroy(rj);
  if (gs) jamd_gms_destroy(gs);
  if (gm) jamd_gmm_destroy(gm);
  if (dn) jamd_dnn_destroy(dn);
  jamd_engine_destroy(e);
  free(text); free(nfr);
  return NULL;
}

/* "0,1,2" or "0-7" or "0-3,0-3" -> device numbers; returns how many (0 = malformed) */
static int parse_devices(const char *spec, int *dev, int cap)
{
  int n = 0;
  const char *p = spec;
  while (*p) {
    char *end;
    long a = strtol(p, &end, 10), b;
    if (end == p || a < 0) return 0;
    b = a;
    if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); if (end == p || b < a) return 0; }
    for (; a <= b; a++) { if (n == cap) return 0; dev[n++] = (int)a; }
    if (*end == ',') end++; else if (*end) return 0;
    p = end;
  }
  return n;
}

int main(int argc, char **argv)
{
  int device = 0, shard_r = 0, shard_n = 1, i, d;
  int devs[64], ndev = 0;
  const char *devspec = NULL;
  char **files = NULL; int nfile = 0, capfile = 0;
  char line[4096];
  FILE *fl;
  devjob jobs[64];
  pthread_t th[64];
  char **out = NULL;

  for (i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-d") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-devices") && i + 1 < argc) devspec = argv[++i];
    else if (!strcmp(argv[i], "-b") && i + 1 < argc) beam = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-bs") && i + 1 < argc) bs = (float)atof(argv[++i]);
    else if (!strcmp(argv[i], "-gprune") && i + 1 < argc) {
      ++i;
      if (!strcmp(argv[i], "safe") && i + 1 < argc) { gprune = JAMD_GPRUNE_SAFE; gnum = atoi(argv[++i]); }
      else if (!strcmp(argv[i], "heu") && i + 1 < argc) { gprune = JAMD_GPRUNE_HEU; gnum = atoi(argv[++i]); }
      else if (!strcmp(argv[i], "beam") && i + 1 < argc) { gprune = JAMD_GPRUNE_BEAM; gnum = atoi(argv[++i]); }
    } else if (!strcmp(argv[i], "-strict")) strict = 1;
    else if (!strcmp(argv[i], "-order") && i + 1 < argc) {          /* exact (default) | fast | strict */
      ++i;
      if (!strcmp(argv[i], "fast")) order = JAMD_ORDER_FAST;
      else if (!strcmp(argv[i], "exact")) order = JAMD_ORDER_EXACT;
      else if (!strcmp(argv[i], "strict")) strict = 1;
      else die("-order exact|fast|strict");
    }
    else if (!strcmp(argv[i], "-shard") && i + 2 < argc) { shard_r = atoi(argv[++i]); shard_n = atoi(argv[++i]); }
    else if (!strcmp(argv[i], "-launch") && i + 1 < argc) launch = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-time")) want_time = 1;
    else if (!strcmp(argv[i], "-am") && i + 1 < argc) am = argv[++i];
    else if (!strcmp(argv[i], "-gms") && i + 1 < argc) gmsp = argv[++i];
    else if (!strcmp(argv[i], "-rej") && i + 1 < argc) rejp = argv[++i];
    else if (!strcmp(argv[i], "-dnnconf") && i + 1 < argc) dnnconf = argv[++i];
    else if (!strcmp(argv[i], "-lex") && i + 1 < argc) lexp = argv[++i];
    else if (!strcmp(argv[i], "-filelist") && i + 1 < argc) list = argv[++i];
    else { fprintf(stderr, "jamd_batch: unknown option %s\n", argv[i]); return 2; }
  }
  if (devspec != NULL) ndev = parse_devices(devspec, devs, 64); else { devs[0] = device; ndev = 1; }
  if ((am == NULL) == (dnnconf == NULL) || (gmsp != NULL && am == NULL) || lexp == NULL || list == NULL || shard_n < 1 || shard_r < 0 || shard_r >= shard_n || launch < 1 || launch > LAUNCH || ndev < 1) {
    fprintf(stderr, "usage: jamd_batch (-am model.blob [-gms selection.blob] | -dnnconf dnn.conf) -lex lexicon.blob -filelist list "
                    "[-d dev | -devices 0,1,..|0-7] [-b beam] [-bs width] [-gprune safe|heu|beam N] [-order exact|fast|strict] [-shard R N] [-launch utterances per launch, 1..512] [-time]\n");
    return 2;
  }
  setenv("GPU_MAX_HW_QUEUES", "16", 0);          /* the upload/scoring stream and the first-pass stream must not share a hardware queue */
  if (jamd_abi_version() != JAMD_ABI_VERSION) { fprintf(stderr, "jamd_batch: ABI mismatch\n"); return 1; }

  if ((fl = fopen(list, "r")) == NULL) { fprintf(stderr, "jamd_batch: cannot open %s\n", list); return 1; }
  while (fgets(line, sizeof(line), fl)) {
    size_t n = strlen(line);
    while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r' || line[n - 1] == ' ')) line[--n] = 0;
    if (n == 0) continue;
    if (nfile == capfile) { capfile = capfile ? 2 * capfile : 64; files = (char **)realloc(files, sizeof(char *) * capfile); }
    files[nfile++] = strdup(line);
  }
  fclose(fl);

  /* utterance u of the list belongs to shard u % (shard_n * ndev); this process holds shards shard_r * ndev + d */
  if (ndev > 1) out = (char **)calloc((size_t)nfile + 1, sizeof(char *));
  for (d = 0; d < ndev; d++) {
    devjob *j = &jobs[d];
    int u;
    j->device = devs[d]; j->nfile = 0; j->out = out;
    j->files = (char **)malloc(sizeof(char *) * ((size_t)nfile + 1));
    j->gidx = (int *)malloc(sizeof(int) * ((size_t)nfile + 1));
    for (u = 0; u < nfile; u++)
      if (u % (shard_n * ndev) == shard_r * ndev + d) { j->files[j->nfile] = files[u]; j->gidx[j->nfile++] = u; }
  }
  if (ndev == 1) run_device(&jobs[0]);
  else {
    for (d = 0; d < ndev; d++) if (pthread_create(&th[d], NULL, run_device, &jobs[d]) != 0) { fprintf(stderr, "jamd_batch: cannot start a device thread\n"); return 1; }
    for (d = 0; d < ndev; d++) pthread_join(th[d], NULL);
    for (i = 0; i < nfile; i++) if (out[i] != NULL) { fputs(out[i], stdout); free(out[i]); }
    free(out);
  }
  for (d = 0; d < ndev; d++) { free(jobs[d].files); free(jobs[d].gidx); }
  for (i = 0; i < nfile; i++) free(files[i]);
  free(files);
  return 0;
}
