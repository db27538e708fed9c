// mmu_abi.hip -- library-wide pieces of the C-ABI (error string, version).
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

thread_local char g_mmu_err[512] = {0};

int mmu_fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_mmu_err, sizeof(g_mmu_err), fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" const char *mmu_last_error(void) { return g_mmu_err; }
extern "C" int mmu_abi_version(void) { return MMU_ABI_VERSION; }
