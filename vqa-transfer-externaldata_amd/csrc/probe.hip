// Measurement probe and trace ranges of libvqahot.so (tracing).
//
// Every launch group of the whole-model entry points sits inside a ProbeScope(label, stream) (vqa_common.h).  A scope
//   * records a pair of HIP events on the group's OWN stream when its label was enabled with vqa_probe_enable -- bench.py
//     reads the per-sample durations back (roofline of the dominant kernel inside the timed region; per-group times of
//     the step in a separate pass), and
//   * opens a roctx range (roctxRangePushA / roctxRangePop) when ranges are on (vqa_roctx_enable(1) or VQA_HOT_ROCTX=1),
//     so that `rocprofv3 --kernel-trace --marker-trace` groups the kernels by phase.  The roctx library is looked up at
//     run time (rocprofiler-sdk's, then roctracer's); without it the ranges are silently off.
// Host code, plus one one-wave kernel: the stream delay that staggers the recurrence's row chains.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "vqa_common.h"

namespace {

struct Samples {
    std::vector<hipEvent_t> ev;      // start / stop pairs, `used` of them recorded
    int used = 0;
};

struct ProbeState {
    std::mutex mu;
    bool on = false;
    bool all = false;                // "*": every label
    int cap = 0;
    std::vector<std::string> order;  // labels in the order given to vqa_probe_enable (first = vqa_probe_read's)
    std::map<std::string, Samples> by_label;
} g;

typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
struct Roctx {
    bool tried = false;
    int on = -1;                     // -1: follow VQA_HOT_ROCTX
    roctx_push_t push = nullptr;
    roctx_pop_t pop = nullptr;
} r;

bool roctx_ready() {
    if (!r.tried) {
        r.tried = true;
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h == nullptr) continue;
            r.push = reinterpret_cast<roctx_push_t>(dlsym(h, "roctxRangePushA"));
            r.pop = reinterpret_cast<roctx_pop_t>(dlsym(h, "roctxRangePop"));
            if (r.push != nullptr && r.pop != nullptr) break;
            r.push = nullptr;
            r.pop = nullptr;
        }
    }
    if (r.push == nullptr) return false;
    if (r.on >= 0) return r.on != 0;
    static const bool env = [] { const char* e = getenv("VQA_HOT_ROCTX"); return e != nullptr && atoi(e) != 0; }();
    return env;
}

}  // namespace

// slot encoding handed back to the scope: bit 0 = a roctx range is open, bits 1.. = 1 + index of the event pair
int vqa_probe_begin(const char* label, hipStream_t st, void** handle) {
    int flags = 0;
    *handle = nullptr;
    if (roctx_ready()) {
        r.push(label);
        flags |= 1;
    }
    if (!g.on) return flags;
    {   // a stream in graph capture: timing events recorded there would become graph nodes; the probe stays out
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
        else if (cs == hipStreamCaptureStatusActive) return flags;
    }
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.on) return flags;
    auto it = g.by_label.find(label);
    if (it == g.by_label.end()) {
        if (!g.all) return flags;
        g.order.push_back(label);
        it = g.by_label.emplace(label, Samples{}).first;
    }
    Samples& s = it->second;
    if (s.used >= g.cap) return flags;
    if ((int)s.ev.size() < 2 * (s.used + 1)) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess) return flags;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return flags; }
        s.ev.push_back(a);
        s.ev.push_back(b);
    }
    (void)hipEventRecord(s.ev[2 * s.used], st);
    *handle = s.ev[2 * s.used + 1];
    s.used++;
    return flags | 2;
}

void vqa_probe_end(int flags, hipStream_t st, void* handle) {
    if ((flags & 2) && handle != nullptr) (void)hipEventRecord(static_cast<hipEvent_t>(handle), st);
    if (flags & 1) r.pop();
}

extern "C" int vqa_probe_disable(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    for (auto& kv : g.by_label)
        for (auto& e : kv.second.ev) (void)hipEventDestroy(e);
    g.by_label.clear();
    g.order.clear();
    g.on = g.all = false;
    g.cap = 0;
    return VQA_OK;
}

extern "C" int vqa_probe_enable(const char* labels, int max_samples) {
    VQA_REQUIRE(labels != nullptr && labels[0] != 0 && max_samples > 0 && max_samples <= 4096, VQA_ERR_ARG);
    vqa_probe_disable();
    std::lock_guard<std::mutex> lk(g.mu);
    std::string all(labels);
    size_t pos = 0;
    while (pos <= all.size()) {
        size_t q = all.find(',', pos);
        if (q == std::string::npos) q = all.size();
        std::string lab = all.substr(pos, q - pos);
        if (lab == "*") g.all = true;
        else if (!lab.empty() && g.by_label.find(lab) == g.by_label.end()) {
            g.order.push_back(lab);
            Samples s;
            s.ev.resize(2 * (size_t)max_samples);
            for (auto& e : s.ev)
                if (hipEventCreate(&e) != hipSuccess) return VQA_ERR_LAUNCH;
            g.by_label.emplace(lab, std::move(s));
        }
        pos = q + 1;
    }
    g.cap = max_samples;
    g.on = true;
    return VQA_OK;
}

extern "C" int vqa_probe_read_label(const char* label, float* ms_out, int capacity, int* n_out) {
    VQA_REQUIRE(ms_out != nullptr && n_out != nullptr && capacity >= 0, VQA_ERR_ARG);
    std::lock_guard<std::mutex> lk(g.mu);
    *n_out = 0;
    std::string key = label != nullptr ? std::string(label) : (g.order.empty() ? std::string() : g.order[0]);
    auto it = g.by_label.find(key);
    if (it == g.by_label.end()) return VQA_OK;
    Samples& s = it->second;
    int n = 0;
    for (int i = 0; i < s.used && n < capacity; ++i) {
        if (hipEventSynchronize(s.ev[2 * i + 1]) != hipSuccess) return VQA_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.ev[2 * i], s.ev[2 * i + 1]) != hipSuccess) return VQA_ERR_LAUNCH;
        ms_out[n++] = ms;
    }
    *n_out = n;
    return VQA_OK;
}

extern "C" int vqa_probe_read(float* ms_out, int capacity, int* n_out) {
    return vqa_probe_read_label(nullptr, ms_out, capacity, n_out);
}

// labels seen so far (enabled ones first, then those discovered under "*"), '\n'-separated; returns the length needed
extern "C" int vqa_probe_labels(char* buf, int capacity) {
    std::lock_guard<std::mutex> lk(g.mu);
    std::string out;
    for (const auto& l : g.order) {
        if (!out.empty()) out += '\n';
        out += l;
    }
    if (buf != nullptr && capacity > 0) {
        strncpy(buf, out.c_str(), (size_t)capacity - 1);
        buf[capacity - 1] = 0;
    }
    return (int)out.size() + 1;
}

// One wave that returns once `ticks` of the 100 MHz real-time counter have passed: a delay ON a stream (it holds one wave
// slot and no other resource), used to start the row chains of the recurrence in anti-phase (csrc/fusion_model.hip).
__global__ __launch_bounds__(64) void delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int vqa_stream_delay_us(float us, void* stream) {
    VQA_REQUIRE(us >= 0.f && us <= 1e5f, VQA_ERR_ARG);
    if (us == 0.f) return VQA_OK;
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       (unsigned long long)(us * 100.f + 0.5f));
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

// Shader-clock sampler: a few single-wave workgroups (dealt round-robin over the XCDs) that sleep beside whatever runs on the
// other streams and, every `ticks` of the 100 MHz real-time counter, note how many shader cycles (s_memtime) went by.  The
// f32 MFMA peak of the roofline assumes 2.4 GHz; under sustained matrix work the chip holds 2.05-2.2 GHz
// (profiles/r3_gru_stamps.txt), and bench.py states the clock its step actually ran at.
__global__ __launch_bounds__(64) void clock_sample_kernel(float* __restrict__ ghz, int n, unsigned long long ticks) {
    for (int i = 0; i < n; ++i) {
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
        unsigned long long r1 = r0;
        while (r1 - r0 < ticks) {
            __builtin_amdgcn_s_sleep(64);
            r1 = __builtin_amdgcn_s_memrealtime();
        }
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) ghz[blockIdx.x * n + i] = (float)((double)(c1 - c0) / (double)(r1 - r0) * 0.1);
    }
}

extern "C" int vqa_clock_sample(float us_per_sample, int n_samples, int n_workgroups, float* ghz_out, void* stream) {
    VQA_REQUIRE(ghz_out != nullptr && us_per_sample >= 1.f && us_per_sample <= 1e5f && n_samples > 0 && n_samples <= 4096 &&
                    n_workgroups > 0 && n_workgroups <= 64 && (double)us_per_sample * n_samples <= 2e6,
                VQA_ERR_ARG);
    hipLaunchKernelGGL(clock_sample_kernel, dim3(n_workgroups), dim3(64), 0, static_cast<hipStream_t>(stream), ghz_out,
                       n_samples, (unsigned long long)(us_per_sample * 100.f + 0.5f));
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_roctx_enable(int on) {
    r.on = on ? 1 : 0;
    return roctx_ready() == (on != 0) ? VQA_OK : VQA_ERR_UNSUPPORTED;
}
