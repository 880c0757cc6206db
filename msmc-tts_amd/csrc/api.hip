// api.hip -- library identity entry points of the C ABI (include/msmc_hip.h).
#include <msmc_rt.hpp>
#include <msmc_hip.h>

extern "C" {
const char* msmc_backend(void) { return MSMC_BACKEND_NAME; }
int msmc_abi_version(void) { return 2; }
}
