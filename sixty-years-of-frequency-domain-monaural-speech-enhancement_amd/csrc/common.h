// Shared host-side declarations for the MI355X speech-enhancement engine.
// gfx950 only: no CUDA shims, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <stdexcept>
#include <string>
#include <vector>
#include "fastmath.h"

namespace se {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define SE_STR2(x) #x
#define SE_STR(x) SE_STR2(x)
#define SE_CHECK(cond, msg)                                                                 \
    do {                                                                                    \
        if (!(cond)) throw ::se::Error(std::string(__FILE__ ":" SE_STR(__LINE__) ": ") + (msg)); \
    } while (0)
#define SE_HIP(expr)                                                                         \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            throw ::se::Error(std::string(__FILE__ ":" SE_STR(__LINE__) ": " #expr ": ") +   \
                              hipGetErrorString(e_));                                       \
    } while (0)

// Bump allocator over one hipMalloc'd slab: activations / scratch are carved at
// create time, nothing is allocated on the decode path (graph-capture safe).
class Arena {
  public:
    Arena() = default;
    ~Arena() { release(); }
    void reserve(size_t bytes) {
        release();
        SE_HIP(hipMalloc(&base_, bytes));
        cap_ = bytes;
        off_ = 0;
    }
    void release() {
        if (base_) (void)hipFree(base_);
        base_ = nullptr;
        cap_ = off_ = 0;
    }
    void reset() {
        off_ = 0;
        // SE_ARENA_POISON=1 (tests): every re-carve fills the slab with NaN patterns, so that a kernel that reads a column
        // nobody wrote shows up deterministically instead of depending on what the previous carve left there
        static const bool poison = getenv("SE_ARENA_POISON") && atoi(getenv("SE_ARENA_POISON")) != 0;
        if (poison && base_ && !measuring_) {
            (void)hipDeviceSynchronize();
            (void)hipMemset(base_, 0xFF, cap_);
            (void)hipDeviceSynchronize();
        }
    }
    // measuring mode: no memory behind the pointers, only the high-water mark is tracked
    void measure_begin() {
        release();
        measuring_ = true;
        cap_ = ~size_t(0) >> 1;
        off_ = 0;
    }
    size_t measure_end() {
        size_t u = off_;
        measuring_ = false;
        cap_ = off_ = 0;
        return u;
    }
    float* alloc_f(size_t n) { return static_cast<float*>(alloc(n * sizeof(float))); }
    void* alloc(size_t bytes) {
        size_t a = (off_ + 255) & ~size_t(255);
        SE_CHECK(a + bytes <= cap_, "arena exhausted: want " + std::to_string(bytes) + " at " +
                                        std::to_string(a) + " of " + std::to_string(cap_));
        off_ = a + bytes;
        return static_cast<char*>(base_) + a;
    }
    size_t used() const { return off_; }
    const void* base() const { return base_; }
    size_t capacity() const { return cap_; }

  private:
    void* base_ = nullptr;
    size_t cap_ = 0, off_ = 0;
    bool measuring_ = false;
};

// true the first time it is called for (flag storage, current device): kernel attributes are per device
inline bool first_on_device(bool (&seen)[64]) {
    int dev = 0;
    SE_HIP(hipGetDevice(&dev));
    SE_CHECK(dev >= 0 && dev < 64, "device ordinal");
    if (seen[dev]) return false;
    seen[dev] = true;
    return true;
}

// Engine-lifetime scratch that grows on first use, one buffer per (purpose, device, stream): work that can be in flight
// at the same time is on different streams (two handles driven asynchronously, one handle's auxiliary streams), work on
// one stream is ordered - so a buffer is never shared by two running kernels.
// A buffer that is outgrown is RETIRED, not freed: a hipGraph captured for a smaller shape keeps the old pointer baked
// into its kernel nodes (cooperative-LSTM exchange / flags, cLN statistics, InstanceNorm partial sums) and the old
// buffer is large enough for that shape, so replaying it after a larger shape has grown the slot stays valid.  Capacity
// at least doubles on growth, so the retired buffers of a slot sum to less than its final size.
struct ScratchPool {
    struct Slot { char* p = nullptr; size_t cap = 0; std::vector<char*> retired; };
    std::map<std::tuple<int, int, hipStream_t>, Slot> slots;
    std::mutex mu;
    int live[64] = {};            // engines alive per device
};
inline ScratchPool& scratch_pool() {
    static ScratchPool pool;
    return pool;
}
inline char* device_scratch(int purpose, size_t need, hipStream_t s) {
    ScratchPool& P = scratch_pool();
    int dev = 0;
    SE_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(P.mu);
    ScratchPool::Slot& sl = P.slots[std::make_tuple(purpose, dev, s)];
    if (need > sl.cap) {
        if (sl.p) sl.retired.push_back(sl.p);
        const size_t cap = need > 2 * sl.cap ? need : 2 * sl.cap;
        SE_HIP(hipMalloc(&sl.p, cap));
        sl.cap = cap;
    }
    return sl.p;
}
// Engine lifetime bookkeeping (se_engine_create / se_engine_destroy): when the LAST engine of a device goes away, the
// device's scratch slots (and the buffers they retired) are freed - a long-lived process that creates and destroys engines
// does not accumulate them, and a recycled stream handle cannot inherit another engine's slot.
inline void scratch_engine_created(int dev) {
    ScratchPool& P = scratch_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (dev >= 0 && dev < 64) ++P.live[dev];
}
inline void scratch_engine_destroyed(int dev) {
    ScratchPool& P = scratch_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (dev < 0 || dev >= 64 || --P.live[dev] > 0) return;
    P.live[dev] = 0;
    for (auto it = P.slots.begin(); it != P.slots.end();) {
        if (std::get<1>(it->first) == dev) {
            if (it->second.p) (void)hipFree(it->second.p);
            for (char* r : it->second.retired) (void)hipFree(r);
            it = P.slots.erase(it);
        } else {
            ++it;
        }
    }
}

// A host copy of one state-dict entry (fp32; int64 buffers are accepted and dropped).
struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

using StateDict = std::map<std::string, HostTensor>;

inline float* to_device(const std::vector<float>& v) {
    float* d = nullptr;
    SE_HIP(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(float)));
    if (!v.empty()) SE_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

// true while the engine decodes a batch as two half-batches side by side (engine.hip): the first pooled auxiliary stream is
// taken, and a model that would fork onto it stays on its own stream (TaylorSENet: two half-batches beat the encoder fork at
// batch 256, 2 441 against 2 376 utt/s)
inline bool& batch_split_active() {
    static thread_local bool v = false;
    return v;
}
// a model's offline fork onto auxiliary streams is captured into a replayed decode's hipGraph (SE_GRAPH_FORK=0: replayed decodes
// stay on one stream, the round-5 rule)
inline bool graph_fork_enabled() {
    static const bool on = !(getenv("SE_GRAPH_FORK") && atoi(getenv("SE_GRAPH_FORK")) == 0);
    return on;
}

}  // namespace se
