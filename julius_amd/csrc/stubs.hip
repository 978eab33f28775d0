// stubs.hip -- entry points declared in include/julius_amd.h whose device
// implementation has not landed yet.  They fail loudly (never fall back to a
// CPU path); each is deleted from this file when its kernel is added.
#include "jamd_internal.h"

#define JAMD_NOT_YET(name)                                            \
  jamd_set_error(name ": not implemented in this build of the engine"); \
  return JAMD_EINVAL

extern "C" {
int jamd_dnn_create(jamd_engine *, const jamd_dnn_desc *, jamd_dnn **) { JAMD_NOT_YET("jamd_dnn_create"); }
void jamd_dnn_destroy(jamd_dnn *) {}
int jamd_dnn_outprob_dev(jamd_dnn *, const float *, int, float *, void *) { JAMD_NOT_YET("jamd_dnn_outprob_dev"); }
int jamd_dnn_outprob_host(jamd_dnn *, const float *, int, float *) { JAMD_NOT_YET("jamd_dnn_outprob_host"); }
}
