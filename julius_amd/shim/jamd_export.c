/*
 * jamd_export -- write the device-side model files from a Julius configuration.
 *
 *   jamd_export [julius options: -C x.jconf | -h hmmdefs -hlist ... -v dict -nlr lm ...] -jamdout PREFIX
 *
 * Loads the acoustic model, dictionary and language model with Julius' OWN readers
 * (libjulius j_config_load_args_new() / j_create_instance_from_jconf(), i.e. rdhmmdef /
 * read_binhmm, voca_load_htkdict, ngram_read_*, init_dfa ...), builds the tree lexicon exactly as
 * the recogniser would (j_final_fusion() inside j_create_instance_from_jconf()), and writes
 *   PREFIX.am    "JAMDGMM1"  flattened GMM-HMM         (jamd_gmm_save; absent for a DNN-HMM, whose
 *                                                       -dnnconf files the engine reads directly)
 *   PREFIX.gms   "JAMDGMM1"  selection model of -gshmm  (jamd_gms_save; only with -gshmm)
 *   PREFIX.rej   "JAMDGMM1"  verification GMMs of -gmm   (jamd_rejgmm_save; only with -gmm)
 *   PREFIX.lex   "JAMDLEX1"  tree lexicon + LM tables   (jamd_lexicon_save)
 * for workers that never link Julius (jamd_gmm_load / jamd_lexicon_load / jamd_dnn_load,
 * julius_amd/host/jamd_batch.c).  Compiled inside the Julius tree like the other shim files.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define JAMD_WITH_LIBJULIUS 1
#include "jamd_flatten.h"

int main(int argc, char **argv)
{
  const char *prefix = NULL;
  char path[4096];
  char **av = (char **)malloc(sizeof(char *) * (size_t)(argc + 1));
  int ac = 0, i, rc = 0;
  Jconf *jconf;
  Recog *recog;
  RecogProcess *r;

  for (i = 0; i < argc; i++) {                       /* take -jamdout out of the Julius arguments */
    if (strcmp(argv[i], "-jamdout") == 0 && i + 1 < argc) { prefix = argv[++i]; continue; }
    av[ac++] = argv[i];
  }
  av[ac] = NULL;
  if (prefix == NULL || strlen(prefix) > sizeof(path) - 8) {
    fprintf(stderr, "usage: jamd_export <julius options> -jamdout PREFIX\n");
    return 2;
  }
  jconf = j_config_load_args_new(ac, av);
  if (jconf == NULL) { fprintf(stderr, "jamd_export: cannot parse the configuration\n"); return 1; }
  recog = j_create_instance_from_jconf(jconf);
  if (recog == NULL) { fprintf(stderr, "jamd_export: cannot load the models\n"); return 1; }
  r = recog->process_list;
  if (r == NULL || r->next != NULL) { fprintf(stderr, "jamd_export: exactly one recognition instance expected\n"); return 1; }

  if (r->am->dnn == NULL) {
    jamd_flat_gmm fg;
    if (jamd_flatten_hmminfo(r->am->hmminfo, &fg) != 0) { fprintf(stderr, "jamd_export: this acoustic model cannot be flattened\n"); return 1; }
    snprintf(path, sizeof(path), "%s.am", prefix);
    rc = jamd_gmm_save(&fg.desc, path);
    jamd_flat_gmm_free(&fg);
    if (rc != JAMD_OK) { fprintf(stderr, "jamd_export: cannot write %s\n", path); return 1; }
    printf("wrote %s\n", path);
    if (r->am->hmmwrk.OP_gshmm != NULL) {           /* gms_init() has run inside j_final_fusion() */
      HMMWork *wrk = &(r->am->hmmwrk);
      jamd_flat_gmm fs;
      const int S = r->am->hmminfo->totalstatenum;
      int *map = (int *)malloc(sizeof(int) * (size_t)S), s;
      if (map == NULL || jamd_flatten_hmminfo(wrk->OP_gshmm, &fs) != 0) { fprintf(stderr, "jamd_export: the selection model cannot be flattened\n"); return 1; }
      for (s = 0; s < S; s++)
        map[s] = (wrk->state2gs[s] >= 0 && wrk->state2gs[s] < wrk->gsset_num) ? wrk->gsset[wrk->state2gs[s]].state->id : -1;
      snprintf(path, sizeof(path), "%s.gms", prefix);
      rc = jamd_gms_save(&fs.desc, map, S, wrk->my_nbest, path);
      jamd_flat_gmm_free(&fs); free(map);
      if (rc != JAMD_OK) { fprintf(stderr, "jamd_export: cannot write %s\n", path); return 1; }
      printf("wrote %s (%d selection states, %d selected per frame)\n", path, wrk->gsset_num, wrk->my_nbest);
    }
  } else printf("DNN-HMM: give the -dnnconf file to jamd_dnn_load() / jamd_batch -dnnconf\n");
  if (recog->gmm != NULL && recog->gc != NULL) {    /* gmm_init() has run inside j_final_fusion() */
    jamd_flat_gmm fv;
    HTK_HMM_Data *hd;
    const int n = recog->gmm->totalhmmnum;
    int *ms = (int *)malloc(sizeof(int) * (size_t)n), k = 0;
    unsigned char *voice = (unsigned char *)malloc((size_t)n);
    const char **names = (const char **)malloc(sizeof(char *) * (size_t)n);
    if (ms == NULL || voice == NULL || names == NULL || jamd_flatten_hmminfo(recog->gmm, &fv) != 0) {
      fprintf(stderr, "jamd_export: the verification GMMs cannot be flattened\n"); return 1;
    }
    for (hd = recog->gmm->start; hd && k < n; hd = hd->next, k++) {       /* gmm.c:593-598, :466-480 */
      ms[k] = hd->s[1]->id; voice[k] = recog->gc->is_voice[k] ? 1 : 0; names[k] = hd->name;
    }
    snprintf(path, sizeof(path), "%s.rej", prefix);
    rc = jamd_rejgmm_save(&fv.desc, ms, n, jconf->reject.gmm_gprune_num, voice, names, path);
    jamd_flat_gmm_free(&fv); free(ms); free(voice); free(names);
    if (rc != JAMD_OK) { fprintf(stderr, "jamd_export: cannot write %s\n", path); return 1; }
    printf("wrote %s (%d verification GMMs, -gmmnum %d)\n", path, n, jconf->reject.gmm_gprune_num);
  }
  {
    jamd_flat_lexicon fl;
    if ((r->am->hmminfo->multipath ? jamd_flatten_lexicon_multipath(r, &fl) : jamd_flatten_lexicon(r, &fl)) != JAMD_OK) { fprintf(stderr, "jamd_export: this lexicon / LM configuration is not covered by the device first pass\n"); return 1; }
    snprintf(path, sizeof(path), "%s.lex", prefix);
    rc = jamd_lexicon_save(&fl.desc, path);
    if (rc == JAMD_OK && r->lmtype == LM_PROB && r->lm != NULL) rc = jamd_lexicon_append_ngram_names(path, r->lm->ngram);
    if (rc == JAMD_OK && r->lmtype == LM_PROB && r->lm != NULL) rc = jamd_lexicon_append_separation(path, r->lm->config->separate_wnum);
    if (rc == JAMD_OK)
      printf("wrote %s (%d nodes, %d words, beam width %d, gprune %s)\n", path, fl.desc.nnode, fl.desc.nword,
             r->trellis_beam_width, r->am->config->gprune_method == GPRUNE_SEL_SAFE ? "safe" : "none/other");
    jamd_flat_lexicon_free(&fl);
    if (rc != JAMD_OK) { fprintf(stderr, "jamd_export: cannot write %s\n", path); return 1; }
  }
  free(av);
  return 0;
}
