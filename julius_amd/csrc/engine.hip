// engine.hip -- engine object, error reporting, device helpers, lookup tables.
//
// The two lookup tables are *model preparation*, not scoring: the reference
// builds them once on the host with libm at start-up (make_log_tbl(),
// libsent/src/phmm/addlog.c:42-57; logistic_table_build(),
// libsent/src/phmm/calc_dnn.c:349-361) and every score afterwards is defined
// in terms of their float contents, so the engine evaluates the same libm
// expressions on the host and uploads the tables.
#include "jamd_internal.h"
#include <cmath>
#include <mutex>

static thread_local char g_err[512] = "";

void jamd_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int jamd_abi_version(void) { return JAMD_ABI_VERSION; }
const char *jamd_last_error(void) { return g_err; }

int jamd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int jamd_engine_create(int device, jamd_engine **out) {
  if (!out) { jamd_set_error("jamd_engine_create: out is NULL"); return JAMD_EINVAL; }
  *out = nullptr;
  // The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and
  // two streams on one queue run their kernels one after the other; the pipelined hosts of this library (scoring
  // stream + first-pass stream, include/julius_amd.h) need them concurrent.  The variable is read when the runtime
  // initialises: set here it takes effect when this is the process's first HIP call; a process that initialised HIP
  // earlier sets it itself (bench.py does, before importing torch).
  // setenv() is not safe against concurrent getenv()/setenv(): once per process, and a multi-threaded host that creates
  // engines from several threads should export the variable itself before it starts them (jamd_batch does).
  static std::once_flag queues_once;
  std::call_once(queues_once, [] { if (!getenv("GPU_MAX_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "16", 0); });
  int n = 0;
  hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess || n <= 0) {
    jamd_set_error("jamd_engine_create: no HIP device available (%s)",
                   err == hipSuccess ? "count is 0" : hipGetErrorString(err));
    return JAMD_ENODEV;
  }
  if (device < 0 || device >= n) {
    jamd_set_error("jamd_engine_create: device %d out of range [0,%d)", device, n);
    return JAMD_EINVAL;
  }
  JAMD_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  JAMD_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    jamd_set_error("jamd_engine_create: device %d is %s; this engine is built for gfx950 only",
                   device, prop.gcnArchName);
    return JAMD_ENODEV;
  }
  jamd_engine *e = new jamd_engine();
  e->device = device;
  e->num_cu = prop.multiProcessorCount;
  JAMD_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));

  // addlog.c:42-57
  std::vector<float> tbl(JAMD_TBLSIZE + 1);
  tbl[JAMD_TBLSIZE] = 0.0f;     // one entry past the reference's table: "no table term" for the pipelined
                                // gathers of the GMM kernels (adding +0.0f changes nothing)
  for (int i = 0; i < JAMD_TBLSIZE; i++) {
    float f = -((float)15 * (float)i / (float)JAMD_TBLSIZE);
    tbl[i] = (float)log(1 + exp(f));
  }
  JAMD_HIP(hipMalloc(&e->d_addlog, sizeof(float) * (JAMD_TBLSIZE + 1)));
  JAMD_HIP(hipMemcpy(e->d_addlog, tbl.data(), sizeof(float) * (JAMD_TBLSIZE + 1), hipMemcpyHostToDevice));
  // calc_dnn.c:349-361
  std::vector<float> sig(JAMD_LOGISTIC_MAX + 1);
  for (int i = 0; i <= JAMD_LOGISTIC_MAX; i++) {
    double x = (double)i / (double)JAMD_LOGISTIC_FACTOR - 8.0;
    sig[i] = (float)(1.0 / (1.0 + exp(-x)));
  }
  JAMD_HIP(hipMalloc(&e->d_logistic, sizeof(float) * (JAMD_LOGISTIC_MAX + 1)));
  JAMD_HIP(hipMemcpy(e->d_logistic, sig.data(), sizeof(float) * (JAMD_LOGISTIC_MAX + 1),
                     hipMemcpyHostToDevice));
  // `tmp < LOG_ADDMIN` (addlog.c:114) compares a float with a double constant;
  // for float tmp that is exactly `tmp < fl_up(LOG_ADDMIN)`.
  float th = (float)JAMD_LOG_ADDMIN;
  if ((double)th < JAMD_LOG_ADDMIN) th = nextafterf(th, INFINITY);
  e->addmin_f = th;
  *out = e;
  return JAMD_OK;
}

void jamd_engine_destroy(jamd_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->d_addlog) (void)hipFree(e->d_addlog);
  if (e->d_logistic) (void)hipFree(e->d_logistic);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int jamd_engine_device(const jamd_engine *e) { return e ? e->device : -1; }

int jamd_engine_sync(jamd_engine *e) {
  if (!e) { jamd_set_error("jamd_engine_sync: NULL engine"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  JAMD_HIP(hipStreamSynchronize(e->stream));
  return JAMD_OK;
}

int jamd_malloc(jamd_engine *e, size_t bytes, void **dev) {
  if (!e || !dev) { jamd_set_error("jamd_malloc: NULL argument"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  JAMD_HIP(hipMalloc(dev, bytes ? bytes : 4));
  return JAMD_OK;
}
int jamd_free(jamd_engine *e, void *dev) {
  if (!e) { jamd_set_error("jamd_free: NULL engine"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  if (dev) JAMD_HIP(hipFree(dev));
  return JAMD_OK;
}
int jamd_host_alloc(jamd_engine *e, size_t bytes, void **host) {
  if (!e || !host) { jamd_set_error("jamd_host_alloc: NULL argument"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  JAMD_HIP(hipHostMalloc(host, bytes ? bytes : 4, hipHostMallocDefault));
  return JAMD_OK;
}
int jamd_host_free(jamd_engine *e, void *host) {
  if (!e) { jamd_set_error("jamd_host_free: NULL engine"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  if (host) JAMD_HIP(hipHostFree(host));
  return JAMD_OK;
}
int jamd_memcpy_h2d(jamd_engine *e, void *dev, const void *host, size_t bytes) {
  if (!e) { jamd_set_error("jamd_memcpy_h2d: NULL engine"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  JAMD_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, e->stream));
  JAMD_HIP(hipStreamSynchronize(e->stream));
  return JAMD_OK;
}
int jamd_stream_create(jamd_engine *e, void **stream) {
  if (!e || !stream) { jamd_set_error("jamd_stream_create: NULL argument"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  hipStream_t st = nullptr;
  JAMD_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *stream = (void *)st;
  return JAMD_OK;
}
int jamd_stream_destroy(jamd_engine *e, void *stream) {
  if (!e) { jamd_set_error("jamd_stream_destroy: NULL engine"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  if (stream) JAMD_HIP(hipStreamDestroy((hipStream_t)stream));
  return JAMD_OK;
}
int jamd_stream_wait(jamd_engine *e, void *waiter, void *signaler) {
  if (!e) { jamd_set_error("jamd_stream_wait: NULL engine"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  hipEvent_t ev = nullptr;
  JAMD_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t rc = hipEventRecord(ev, jamd_stream(e, signaler));
  if (rc == hipSuccess) rc = hipStreamWaitEvent(jamd_stream(e, waiter), ev, 0);
  (void)hipEventDestroy(ev);                       // released once the recorded work has completed
  if (rc != hipSuccess) { jamd_set_error("jamd_stream_wait: %s", hipGetErrorString(rc)); return JAMD_ENODEV; }
  return JAMD_OK;
}
int jamd_stream_sync(jamd_engine *e, void *stream) {
  if (!e) { jamd_set_error("jamd_stream_sync: NULL engine"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  JAMD_HIP(hipStreamSynchronize(jamd_stream(e, stream)));
  return JAMD_OK;
}
int jamd_memcpy_h2d_async(jamd_engine *e, void *dev, const void *host, size_t bytes, void *stream) {
  if (!e) { jamd_set_error("jamd_memcpy_h2d_async: NULL engine"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  JAMD_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, jamd_stream(e, stream)));
  return JAMD_OK;
}
int jamd_memcpy_d2h(jamd_engine *e, void *host, const void *dev, size_t bytes) {
  if (!e) { jamd_set_error("jamd_memcpy_d2h: NULL engine"); return JAMD_EINVAL; }
  JAMD_HIP(hipSetDevice(e->device));
  JAMD_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, e->stream));
  JAMD_HIP(hipStreamSynchronize(e->stream));
  return JAMD_OK;
}

}  // extern "C"
