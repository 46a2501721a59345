// Device memory of the library's long-lived buffers (activation arenas, statistics, integrator state, parameter stores, job tables):
// every such buffer comes from dev_alloc() so that ONE switch turns all of them into poisoned, fenced allocations.
//
// Why (round 4): three failures of two model replicas running side by side (a NaN / non-reproducible result twice, a memory access fault
// once) could not be told apart from the records -- a kernel that reads a slot nobody wrote, or a few bytes past the end of its buffer,
// gets finite leftovers in one process layout and NaN bit patterns or an unmapped page in another.  Under FLOCODER_AMD_POISON=1
// (or fc_debug_set_poison(1)) every buffer is
//   * filled with 0xFFFFFFFF words (a NaN as fp32, -1 as an index) before it is handed out: a value that is read before it is written
//     poisons the result deterministically -- also where it is multiplied by zero afterwards;
//   * surrounded by two 64 KiB fences of the same pattern: a read past either end (a prefetch one chunk ahead, a 16-byte load at a
//     ragged tail) stays inside mapped memory and returns NaN, a WRITE past either end is found by fc_debug_poison_check().
// Without the switch dev_alloc is hipMalloc.  Test infrastructure of the product library (tests/test_gpu_poison.py); not a fallback.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace fc {

namespace {
constexpr size_t kFence = 64 * 1024;
constexpr unsigned kPattern = 0xFFFFFFFFu;
struct Rec { void* base; size_t bytes, padded; std::string tag; };
std::mutex g_mu;
std::map<void*, Rec> g_live;      // user pointer -> record (poison mode only)
int g_poison = -1;                // -1: ask the environment at first use

bool poison_on() {
    if (g_poison < 0) {
        const char* e = std::getenv("FLOCODER_AMD_POISON");
        g_poison = (e && *e && std::strcmp(e, "0") != 0) ? 1 : 0;
    }
    return g_poison == 1;
}
}  // namespace

int dev_alloc(void** out, size_t bytes, const char* tag) {
    if (!out) return fail(FC_E_ARG, "dev_alloc: null argument");
    if (bytes == 0) bytes = 4;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!poison_on()) {
        FC_HIP(hipMalloc(out, bytes));
        return FC_OK;
    }
    const size_t padded = (bytes + 255) & ~(size_t)255;
    void* base = nullptr;
    FC_HIP(hipMalloc(&base, padded + 2 * kFence));
    FC_HIP(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(base), (int)kPattern, (padded + 2 * kFence) / 4));
    FC_HIP(hipDeviceSynchronize());
    void* user = static_cast<char*>(base) + kFence;
    g_live[user] = Rec{base, bytes, padded, tag ? tag : ""};
    *out = user;
    return FC_OK;
}

void dev_free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) { (void)hipFree(p); return; }   // allocated while the switch was off
    (void)hipFree(it->second.base);
    g_live.erase(it);
}

}  // namespace fc

extern "C" {

int fc_debug_set_poison(int on) {
    std::lock_guard<std::mutex> lk(fc::g_mu);
    fc::g_poison = on ? 1 : 0;
    return FC_OK;
}

// Looks at both fences (and the rounding slack behind the last byte) of every live poisoned buffer.  *corrupted = buffers with a
// changed fence word, *live = buffers looked at; fc_last_error() names the first few.  Synchronises the device.
int fc_debug_poison_check(int* corrupted, int* live) {
    if (!corrupted) return fc::fail(FC_E_ARG, "fc_debug_poison_check: null argument");
    std::lock_guard<std::mutex> lk(fc::g_mu);
    FC_HIP(hipDeviceSynchronize());
    int bad = 0;
    std::string report;
    std::vector<unsigned> host;
    for (const auto& kv : fc::g_live) {
        const fc::Rec& r = kv.second;
        const size_t tail = fc::kFence + (r.padded - ((r.bytes + 3) & ~(size_t)3));
        const char* front = static_cast<const char*>(r.base);
        const char* back = static_cast<const char*>(kv.first) + ((r.bytes + 3) & ~(size_t)3);
        long first_front = -1, first_back = -1;
        host.resize(fc::kFence / 4);
        FC_HIP(hipMemcpy(host.data(), front, fc::kFence, hipMemcpyDeviceToHost));
        for (size_t i = host.size(); i-- > 0;) if (host[i] != fc::kPattern) { first_front = (long)(fc::kFence - 4 * i); break; }   // bytes in front of the buffer
        host.resize(tail / 4);
        FC_HIP(hipMemcpy(host.data(), back, tail, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < host.size(); ++i) if (host[i] != fc::kPattern) { first_back = (long)(4 * i); break; }                 // bytes behind its end
        if (first_front >= 0 || first_back >= 0) {
            if (bad < 8)
                report += " [" + r.tag + ": " + std::to_string(r.bytes) + " bytes" + (first_front >= 0 ? ", written " + std::to_string(first_front) + " bytes in front" : "") +
                          (first_back >= 0 ? ", written " + std::to_string(first_back) + " bytes behind the end" : "") + "]";
            ++bad;
        }
    }
    *corrupted = bad;
    if (live) *live = (int)fc::g_live.size();
    if (bad) fc::set_error("poison check: " + std::to_string(bad) + " buffer(s) written out of bounds:" + report);
    return FC_OK;
}

}  // extern "C"
