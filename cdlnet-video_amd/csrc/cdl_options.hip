// Process-wide switches of libcdlnet_hip.so and the per-device bookkeeping the launchers need.
//
// The compute entry points keep no state of their own.  What is process-wide lives here, behind locks:
//   * experiment / test switches, read from the environment ONCE (first use) into an immutable snapshot;
//     cdl_options_reload() re-reads them (tests and tools flip a variable, then call it);
//   * per-device facts: compute-unit count, and which kernels already had their dynamic-LDS limit raised
//     and to how many bytes on which device (hipFuncSetAttribute is per device: a second GPU in the same process needs its own call).
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "cdl_common.h"

namespace {

std::atomic<const cdl_options *> g_opts{nullptr};
std::mutex g_opts_mutex;

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
int env_flag_off(const char *name)          // 1 unless the variable starts with '0'
{
    const char *e = getenv(name);
    return (e && e[0] == '0') ? 0 : 1;
}

const cdl_options *load_options()
{
    cdl_options *o = new cdl_options;       // snapshots are never freed: a reader may still hold the old one
    o->mfma_analysis = env_flag_off("CDL_MFMA_ANALYSIS");
    o->mfma_synthesis = env_flag_off("CDL_MFMA_SYNTHESIS");
    o->mfma_wgrad = env_flag_off("CDL_MFMA_WGRAD");
    o->mfma_dense = env_flag_off("CDL_MFMA_DENSE");
    o->no_tiled = getenv("CDL_NO_TILED") ? 1 : 0;
    o->no_pipelined_synthesis = getenv("CDL_NO_PIPELINED_SYNTHESIS") ? 1 : 0;
    o->fused_snake = env_flag_off("CDL_FUSED_SNAKE");
    o->fused_grid = env_int("CDL_FUSED_GRID", 0);
    o->fused_da = env_flag_off("CDL_FUSED_DA");
    o->scalar_assemble = env_int("CDL_SCALAR_ASSEMBLE", 0) ? 1 : 0;
    o->fusedg_strip = env_int("CDL_FUSEDG_STRIP", 0) ? 1 : 0;
    o->fusedg_bwd_prec = env_int("CDL_FUSEDG_PREC", -1);
#ifdef CDL_ABLATE
    o->fused_debug = env_int("CDL_FUSED_DEBUG", 0);
    o->dense_debug = env_int("CDL_DENSE_DEBUG", 0);
#else
    o->fused_debug = o->dense_debug = 0;
#endif
    return o;
}

thread_local int t_exact_fp32 = 0;          // cdl_set_exact_fp32: per host thread, off the snapshot

constexpr int MAX_DEV = 64;
std::atomic<int> g_cus[MAX_DEV];
std::mutex g_attr_mutex;
std::map<std::pair<int, const void *>, int> g_attr_bytes;     // largest limit set so far per (device, kernel)

}  // namespace

const cdl_options &cdl_opts()
{
    const cdl_options *o = g_opts.load(std::memory_order_acquire);
    if (!o) {
        std::lock_guard<std::mutex> lk(g_opts_mutex);
        o = g_opts.load(std::memory_order_acquire);
        if (!o) {
            o = load_options();
            g_opts.store(o, std::memory_order_release);
        }
    }
    return *o;
}

bool cdl_exact_fp32() { return t_exact_fp32 != 0; }

extern "C" int cdl_set_exact_fp32(int on)
{
    const int prev = t_exact_fp32;
    t_exact_fp32 = on ? 1 : 0;
    return prev;
}

extern "C" int cdl_options_reload(void)
{
    std::lock_guard<std::mutex> lk(g_opts_mutex);
    g_opts.store(load_options(), std::memory_order_release);
    return 0;
}

int cdl_current_device()
{
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}

int cdl_cu_count()
{
    const int dev = cdl_current_device();
    if (dev < 0 || dev >= MAX_DEV) return 256;
    int n = g_cus[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        g_cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

int cdl_ensure_dynamic_lds(const void *kernel, int bytes)
{
    const int dev = cdl_current_device();
    std::lock_guard<std::mutex> lk(g_attr_mutex);
    auto it = g_attr_bytes.find({dev, kernel});
    if (it != g_attr_bytes.end() && it->second >= bytes) return 0;
    // one template instantiation serves several LDS carvings (e.g. k_stage_g at M = 48 and M = 64): the limit
    // must follow the LARGEST request, not the first
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return -(int)e;
    g_attr_bytes[{dev, kernel}] = bytes;
    return 0;
}
