// Version / error strings of libdvf_hip.so.
#include "dvf_common.h"

DvfPlanLog &dvf_plan_log() {
    static thread_local DvfPlanLog log{};
    return log;
}

extern "C" {

int dvf_version(void) { return 200; }

int dvf_conv2d_last_plans(int *out, int max_ints) {
    if (!out || max_ints < 0) return DVF_ERR_INVALID_ARG;
    const DvfPlanLog &l = dvf_plan_log();
    int n = 0;
    for (int r = 0; r < l.n && n + DVF_PLAN_INTS <= max_ints; ++r)
        for (int x = 0; x < DVF_PLAN_INTS; ++x) out[n++] = l.rec[r][x];
    return n;
}

int dvf_build_has_tuning(void) {
#ifdef DVF_TUNING
    return 1;
#else
    return 0;
#endif
}

const char *dvf_error_string(int code) {
    switch (code) {
        case DVF_OK: return "ok";
        case DVF_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size or unsupported shape)";
        case DVF_ERR_LAUNCH: return "HIP launch / runtime error";
        case DVF_ERR_UNSUPPORTED: return "configuration not supported by this build";
        default: return "unknown error";
    }
}

}  // extern "C"
