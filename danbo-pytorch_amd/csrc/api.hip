// Host-only entry points of libdanbo_hip.so.
#include <string.h>
#include "common.hpp"

extern "C" int danbo_abi_version(void) { return 9; }

extern "C" int danbo_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
}
