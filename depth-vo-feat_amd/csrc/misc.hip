// Version / error strings of libdvf_hip.so.
#include "dvf_common.h"

extern "C" {

int dvf_version(void) { return 100; }

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
