// Library-level entry points of include/nic_rollout.h.
#include "nic_common.h"

namespace nic {
char* last_error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
const char** last_kernel_slot() {
    static thread_local const char* name = "";
    return &name;
}
char* last_kernel_buffer() {
    static thread_local char buf[160] = {0};
    return buf;
}
}  // namespace nic

extern "C" {

const char* nic_last_kernel(void) { return *nic::last_kernel_slot(); }

int nic_abi_version(void) { return NIC_ABI_VERSION; }

// identity of the sources this library was built from (neural_inventory_control_amd/build.py::source_id passes it in)
#ifndef NIC_BUILD_ID
#define NIC_BUILD_ID "unknown"
#endif
const char* nic_build_id(void) { return NIC_BUILD_ID; }

const char* nic_last_error(void) { return nic::last_error_buffer(); }

int nic_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
}
