// api.hip -- library identity entry points of the C ABI (include/msmc_hip.h).
#include <stdlib.h>
#include <msmc_rt.hpp>
#include <msmc_hip.h>

extern "C" {
const char* msmc_backend(void) { return MSMC_BACKEND_NAME; }
int msmc_abi_version(void) { return 2; }

// ---- streams of the library's own -------------------------------------------------------------------------
int msmc_stream_create(msmc_stream* out) {
    if (!out) return MSMC_E_SHAPE;
    return msmc_rt_stream_create((void**)out);
}
int msmc_stream_destroy(msmc_stream stream) { return msmc_rt_stream_destroy((void*)stream); }

// ---- per-launch profiling log (process-wide) ---------------------------------------------------------------
void msmc_prof_enable(int on) {
    MsmcProfLog& L = msmc_prof_log;
    if (on && !L.rec) L.rec = (MsmcProfRec*)malloc(sizeof(MsmcProfRec) * MSMC_PROF_MAX);
    if (on) msmc_prof_reset_impl();
    L.on = on && L.rec;
}
int msmc_prof_count(void) { return msmc_prof_used(); }
int msmc_prof_read(int i, char* name, int cap, float* ms) { return msmc_prof_read_impl(i, name, cap, ms); }
}
