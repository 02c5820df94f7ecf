// C-ABI + host-side engine of the MI355X WildlifeMapper inference path.
// See include/wm_hip.h for the contract.  No torch types, no CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <tuple>
#include <mutex>
#include <set>
#include <string>
#include <algorithm>
#include <array>
#include <vector>

#include "../../include/wm_hip.h"
#include "attn16.h"
#include "attn_glob8.h"
#include "dec_kernels.h"
#include "fft_kernels.h"
#include "gemm16.h"
#include "gemm16_v2.h"
#include "gemm16_v3.h"
#include "gemm16_v5.h"
#include "gemm32.h"
#include "gemm8.h"
#include "misc_kernels.h"
#include "wm_common.h"

// Dev instrumentation (in-kernel timelines of the GEMMs: tools/gemm_bench.py with WM_GEMM_DBG / WM_GEMM8_DBG /
// WM_GEMM8_DBG) is compiled only with -DWM_DEV_TIMELINE=1 (tools/build_dev.sh); the product library carries neither
// the instrumented kernel instances nor their environment switches.
#ifndef WM_DEV_TIMELINE
#define WM_DEV_TIMELINE 0
#endif

using namespace wm;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define WM_TRY(expr)            \
    do {                        \
        int _r = (expr);        \
        if (_r != 0) return _r; \
    } while (0)

extern "C" const char* wm_last_error(void) { return g_err; }
extern "C" int wm_abi_version(void) { return WM_ABI_VERSION; }

// ---------------------------------------------------------------------------
// per-device launcher state (a process may drive several devices, one handle each)
// ---------------------------------------------------------------------------
static std::mutex g_dev_mu;

// hipFuncAttributeMaxDynamicSharedMemorySize is per (function, device)
static int set_max_lds(const void* fn, int bytes) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    static std::set<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (done.count({fn, dev})) return 0;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({fn, dev});
    return 0;
}

// Scratch memory of the handle-less single-op entry points (tests, tools), one buffer per (device, stream, use): a stream's launches
// are ordered, two streams never share a buffer.  Grown by free + malloc (hipFree synchronises the device).  Handles own their own.
static int op_scratch(hipStream_t s, int use, size_t bytes, void** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    struct Buf { void* p = nullptr; size_t cap = 0; };
    static std::map<std::tuple<int, hipStream_t, int>, Buf> bufs;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    Buf& b = bufs[std::make_tuple(dev, s, use)];
    if (b.cap < bytes) {
        if (b.p) hipFree(b.p);
        b.p = nullptr; b.cap = 0;
        HIP_TRY(hipMalloc(&b.p, bytes));
        b.cap = bytes;
    }
    *out = b.p;
    return 0;
}

static int num_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    static std::map<int, int> cus;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    cus[dev] = n;
    return n;
}

// 256 B of zeros per device: source of out-of-image taps of the implicit-GEMM conv
static int zero_page_for_device(const uint16_t** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    static std::map<int, uint16_t*> pages;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = pages.find(dev);
    if (it == pages.end()) {
        uint16_t* p = nullptr;
        HIP_TRY(hipMalloc((void**)&p, 256));
        HIP_TRY(hipMemset(p, 0, 256));
        it = pages.emplace(dev, p).first;
    }
    *out = it->second;
    return 0;
}

// which GEMM kernel instance each launch took (wm_debug_gemm_variant_counts): the tests assert on it, so that a
// change of the dispatch heuristic cannot silently leave an instance without a value check
static std::atomic<int64_t> g_variant_count[WM_GEMM_VARIANT_COUNT];
static inline void count_variant(int v) { g_variant_count[v].fetch_add(1, std::memory_order_relaxed); }

extern "C" int wm_debug_gemm_variant_counts(int64_t* out, int n) {
    if (!out || n < WM_GEMM_VARIANT_COUNT) return fail("wm_debug_gemm_variant_counts: need room for %d counters", WM_GEMM_VARIANT_COUNT);
    for (int i = 0; i < WM_GEMM_VARIANT_COUNT; ++i) out[i] = g_variant_count[i].load(std::memory_order_relaxed);
    return 0;
}
extern "C" int wm_debug_reset_gemm_variant_counts(void) {
    for (auto& c : g_variant_count) c.store(0, std::memory_order_relaxed);
    return 0;
}

// ---------------------------------------------------------------------------
// host-side 16-bit conversion (round to nearest even), used by the weight packer
// ---------------------------------------------------------------------------
static inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline uint16_t f32_to_f16_host(float f) {
    _Float16 h = (_Float16)fminf(fmaxf(f, -65504.f), 65504.f);
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
}

// f32 -> OCP e4m3fn (bias 7, max 448, no infinity, 0x7f = NaN), round to nearest even, saturating; weight packer of WM_PREC_FP8
static inline uint8_t f32_to_e4m3_host(float f) {
    if (f != f) return 0x7f;
    const uint8_t sign = std::signbit(f) ? 0x80 : 0;
    float a = fabsf(f);
    if (a > 448.0f) a = 448.0f;
    if (a == 0.0f) return sign;
    int e;
    (void)frexpf(a, &e);                                           // a = m 2^e, m in [0.5, 1): floor(log2 a) = e - 1
    int fl = e - 1;
    if (fl < -6) fl = -6;                                          // subnormals share the quantum 2^-9
    const float q = ldexpf(1.0f, fl - 3);
    float v = nearbyintf(a / q) * q;                               // exact scaling; default rounding mode = nearest even
    if (v == 0.0f) return sign;
    if (v > 448.0f) v = 448.0f;
    if (v < ldexpf(1.0f, -6)) return sign | (uint8_t)(int)(v / ldexpf(1.0f, -9));
    (void)frexpf(v, &e);
    const int ex = e - 1;
    const int man = (int)((v / ldexpf(1.0f, ex) - 1.0f) * 8.0f);
    return sign | (uint8_t)(((ex + 7) << 3) | man);
}

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
namespace {

constexpr int T = 4096;          // tokens per tile (64 x 64)
constexpr int GRID = 64;
constexpr int HFC = 1024;        // HFC adaptor width (image_encoder.py:65-87)
constexpr int HFC_HEADS = 8;
constexpr int OUTC = 256;        // neck / decoder width
constexpr int NQ = WM_NUM_QUERIES;
constexpr int DEC_MLP = 2048;

struct HostW {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct EvPair {
    hipEvent_t a, b;
    int kclass;
    double flops, bytes;
};

struct Profiler {
    bool on = false;
    std::vector<EvPair> used;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    wm_kclass_stat acc[WM_KCLASS_COUNT] = {};
};

}  // namespace

struct wm_handle {
    wm_config cfg{};
    int fp8_gemms = WM_FP8_ALL;             // fp8 mode: which of a block's GEMMs run e4m3 (wm_config.fp8_gemms, 0 = all)
    int fp8_bf16_tail = 0, fp8_bf16_head = 0;  // fp8 mode: the first / last blocks that stay bf16 (env WM_FP8_BF16_HEAD / _TAIL, default 0)
    int fp16_tail = 0;      // bf16 mode: the last fp16_tail transformer blocks use fp16 operands (parity margin dial; default 0)
    int device = 0;
    int D = 0, depth = 0, heads = 0, hd = 0, prec = 0, maxB = 0;
    bool is_global[64] = {};
    bool finalized = false, enc_ready = false, dec_ready = false;
    std::map<std::string, std::vector<int64_t>> expected;   // name -> shape
    std::map<std::string, HostW> staged;
    std::map<std::string, uint16_t*> w16;
    // folded LayerNorm (WM_CFG_FOLD_LN): per consumer GEMM weight name: gamma (.) W in LDS-image order, c1, c2; per-row partial
    // statistics of the residual stream [maxB * 4096][<= 4][2]
    std::map<std::string, uint16_t*> wfold;
    std::map<std::string, float*> fold_c1, fold_c2;
    std::map<std::string, float*> wsrc32;   // fp32 device copies of the weights that get folded (qkv, lin1 of every block): gamma (.) W is rounded once
    float* fold_stats = nullptr;
    // split stream (gemm16_v5.h "Split stream"): xn16 = hi plane, lo16 = lo plane, both LDS-image order; `split`: used wherever the
    // residual GEMMs of a folded block run the 256-row-tile kernel (WM_STREAM_SPLIT=0 keeps the fp32 stream: A/B runs).
    // overflow: host-pinned, device-visible word the stream's producers set when an fp16 hi plane clamps (wm_stream_overflow).
    uint16_t* lo16 = nullptr;
    bool split = false;
    // fp8 blocks (round 4): the stream as two planes of rows (x16last = hi bf16, lo16; gemm8.h PLANES) between the e4m3 residual GEMMs, so the
    // LayerNorm-to-e4m3 pass reads 2 bytes per element (gemm8.h PLANES); WM_FP8_ROWS=0 keeps the fp32 stream (A/B runs)
    bool rows8 = true;
    bool fold_from16 = false;               // WM_FOLD_FROM16=1 (A/B runs): gamma (.) W from the 16-bit weight, rounded twice (round 3's form)
    int* overflow = nullptr;
    bool fold = false, fold_bf16 = false;   // WM_CFG_FOLD_LN: fp16-operand blocks; WM_CFG_FOLD_LN_BF16: bf16-operand blocks too
    std::map<std::string, uint16_t*> w16p;  // the same weights in LDS-image order (gemm16_v5.h "Operand layout"), for the 256-row-tile kernels
    std::map<std::string, uint8_t*> w8k;    // qkv / lin1 of the fp8 blocks again with the K columns at wm::plane_pos (operand = layernorm_plane_fp8_kernel's output)
    std::map<std::string, uint8_t*> w8;     // WM_PREC_FP8: e4m3 weights of the blocks' GEMMs; their per-channel scales live in w32[name + ".wscale"]
    uint8_t* ao8 = nullptr;                 // attention output as e4m3 (A operand of proj)
    std::map<std::string, float*> w32;
    std::vector<void*> allocs;
    Profiler prof;
    std::map<const float*, std::pair<uint16_t*, uint16_t*>> w32x3;   // decoder weights as fp16 (hi, lo) planes of W * 2^6 (gemm32.h gemm32x3_kernel), by fp32 copy
    std::map<std::pair<const float*, int>, uint16_t*> bias16;   // qkv biases rounded to a 16-bit operand type (window attention's padded tokens), by (fp32 copy, type)
    float* mha_part = nullptr;              // token -> image attention: per key chunk partial softmaxes (launch_mha32)
    size_t mha_part_cap = 0;
    bool row_major = false;                 // WM_ROW_MAJOR_OPERANDS=1 (A/B runs): no operand in LDS-image order
    bool sat_on = false;                    // wm_debug_saturation_enable
    unsigned long long* sat_counts = nullptr;   // [WM_SAT_COUNT] device counters
    int tap_which = -2;
    float* tap_buf = nullptr;

    // workspace (device)
    float *resid = nullptr, *tokbase = nullptr;
    uint16_t *xn16 = nullptr, *ao16 = nullptr, *qkv16 = nullptr, *hid16 = nullptr;
    uint16_t *p16 = nullptr, *h16 = nullptr, *he16 = nullptr, *hp16 = nullptr, *pt16 = nullptr, *q16 = nullptr,
             *kv16 = nullptr, *aoh16 = nullptr, *y1n16 = nullptr, *h1_16 = nullptr, *y2_16 = nullptr, *y2t16 = nullptr;
    float *pt32 = nullptr, *y1 = nullptr, *y1n32 = nullptr, *z32 = nullptr;
    float *n1 = nullptr, *n2 = nullptr, *emb_nhwc = nullptr, *emb_nchw = nullptr;
    uint16_t *n1n16 = nullptr, *x16last = nullptr;
    float *dkeys = nullptr, *dk_a = nullptr, *dk_b = nullptr, *dk_c = nullptr;      // [B*T,256],[B*T,128] x3
    float *dq = nullptr, *dt_q = nullptr, *dt_k = nullptr, *dt_v = nullptr, *dt_att = nullptr, *dt_hid = nullptr,
          *dt_h1 = nullptr, *dt_h2 = nullptr;
    float *logits = nullptr, *boxes = nullptr, *hfc = nullptr, *tsz_default = nullptr;
    float2 *fftR = nullptr, *fft_tw = nullptr;
    float* kpe = nullptr;           // dense PE, token-major [T,256]
    wm_box_record* records = nullptr;
};

namespace {

template <class P>
int dalloc(wm_handle* h, P** out, size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    h->allocs.push_back(p);
    *out = (P*)p;
    return 0;
}

// -------- profiled launch bracket --------
struct Bracket {
    wm_handle* h;
    hipStream_t s;
    int idx = -1;
    Bracket(wm_handle* h_, hipStream_t s_, int kclass, double flops, double bytes) : h(h_), s(s_) {
        if (!h || !h->prof.on) return;
        Profiler& p = h->prof;
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (!p.pool.empty()) { ev = p.pool.back(); p.pool.pop_back(); }
        else { hipEventCreate(&ev.first); hipEventCreate(&ev.second); }
        hipEventRecord(ev.first, s);
        p.used.push_back(EvPair{ev.first, ev.second, kclass, flops, bytes});
        idx = (int)p.used.size() - 1;
    }
    ~Bracket() {
        if (idx >= 0) hipEventRecord(h->prof.used[idx].b, s);
    }
};

int prof_collect(wm_handle* h) {
    Profiler& p = h->prof;
    for (auto& e : p.used) {
        HIP_TRY(hipEventSynchronize(e.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
        p.acc[e.kclass].launches += 1;
        p.acc[e.kclass].ms += ms;
        p.acc[e.kclass].flops += e.flops;
        p.acc[e.kclass].bytes += e.bytes;
        p.pool.push_back({e.a, e.b});
    }
    p.used.clear();
    return 0;
}

// -------- launchers --------
template <class T16>
int launch_gemm16_t(wm_handle* h, hipStream_t s, const Gemm16Args& a) {
    WM_TRY(set_max_lds((const void*)gemm16_kernel<T16>, G16_LDS_BYTES));
    count_variant(WM_GEMM_V1_128);
    const int grid = (a.M / G16_BM) * (a.N / G16_BN);
    Bracket br(h, s, WM_KCLASS_GEMM16, 2.0 * a.M * (double)a.N * a.K,
               2.0 * ((double)a.M * a.K + (double)a.N * a.K) + (a.out32 ? 4.0 : 0.0) * a.M * a.N + (a.out16 ? 2.0 : 0.0) * a.M * a.N +
                   (a.residual ? 4.0 * (double)(a.res_mod > 0 ? a.res_mod : a.M) * a.N : 0.0));
    hipLaunchKernelGGL(gemm16_kernel<T16>, dim3(grid), dim3(256), G16_LDS_BYTES, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T16, int BN>
int launch_gemm16v2_t(wm_handle* h, hipStream_t s, const Gemm16Args& a) {
    WM_TRY(set_max_lds((const void*)gemm16v2_kernel<T16, BN>, G2<BN>::LDS));
    count_variant(BN == 160 ? WM_GEMM_V2_160 : WM_GEMM_V2_128);
    const int grid = (a.M / 256) * (a.N / BN);
    Bracket br(h, s, WM_KCLASS_GEMM16, 2.0 * a.M * (double)a.N * a.K,
               2.0 * ((double)a.M * a.K + (double)a.N * a.K) + (a.out32 ? 4.0 : 0.0) * a.M * a.N + (a.out16 ? 2.0 : 0.0) * a.M * a.N +
                   (a.residual ? 4.0 * (double)(a.res_mod > 0 ? a.res_mod : a.M) * a.N : 0.0));
    hipLaunchKernelGGL((gemm16v2_kernel<T16, BN>), dim3(grid), dim3(512), G2<BN>::LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// dev build only: tile-order A/B (WM_GEMM_GROUP_M = row tiles per group of the grouped order, every 256-row-tile instance)
static inline void dev_group_m(Gemm16Args& a) {
#if WM_DEV_TIMELINE
    if (const char* e = getenv("WM_GEMM_GROUP_M")) { if (atoi(e) > 0) a.group_m = atoi(e); }
#else
    (void)a;
#endif
}

template <class T16, int BN, int NSLOT = 3>
int launch_gemm16v5_t(wm_handle* h, hipStream_t s, const Gemm16Args& a_in) {
    using G = G3<BN, 4>;
    constexpr int LDS = NSLOT * G::STAGE + 32 * BN * 4;        // ring + the first residual landing buffer
    static_assert(LDS <= 160 * 1024, "LDS");
    WM_TRY(set_max_lds((const void*)gemm16v5_kernel<T16, BN, NSLOT>, LDS));
    count_variant(BN == 320 ? (a_in.residual ? WM_GEMM_V5_320_RES : WM_GEMM_V5_320) : (a_in.residual ? WM_GEMM_V5_256_RES : WM_GEMM_V5_256));
    const int grid = (a_in.M / 256) * (a_in.N / BN);
    Gemm16Args a = a_in;
    dev_group_m(a);
    Bracket br(h, s, WM_KCLASS_GEMM16, 2.0 * a.M * (double)a.N * a.K,
               2.0 * ((double)a.M * a.K + (double)a.N * a.K) + (a.out32 ? 4.0 : 0.0) * a.M * a.N + (a.out16 ? 2.0 : 0.0) * a.M * a.N +
                   (a.residual ? 4.0 * (double)(a.res_mod > 0 ? a.res_mod : a.M) * a.N : 0.0));
#if WM_DEV_TIMELINE
    if constexpr (BN == 320 && NSLOT == 3) {
        static const bool dbg = getenv("WM_GEMM_DBG") != nullptr;          // dev: in-kernel interval timing of workgroup 0
        static int dbg_count = 0;          // instrument the 1st and, after a run of back-to-back launches, the 31st
        if (dbg) ++dbg_count;
        if (dbg && (dbg_count == 1 || dbg_count == 31)) {
            static unsigned* buf = nullptr;
            const size_t bytes = 512 + (size_t)grid * 40;
            if (!buf) HIP_TRY(hipMalloc((void**)&buf, 512 + 8192 * 40));
            if (grid > 8192) return fail("dbg grid");
            HIP_TRY(hipMemsetAsync(buf, 0, bytes, s));
            Gemm16Args d = a;
            d.zero_page = (const u16*)buf;
            WM_TRY(set_max_lds((const void*)gemm16v5_kernel<T16, BN, NSLOT, true>, LDS));
            hipLaunchKernelGGL((gemm16v5_kernel<T16, BN, NSLOT, true>), dim3(grid), dim3(512), LDS, s, d);
            HIP_TRY(hipStreamSynchronize(s));
            std::vector<unsigned char> hbuf(bytes);
            HIP_TRY(hipMemcpy(hbuf.data(), buf, bytes, hipMemcpyDeviceToHost));
            const unsigned* hb = (const unsigned*)hbuf.data();
            fprintf(stderr, "[gemm16v5 dbg] M=%d N=%d K=%d  marks relative to group-0 mark 0 of step 8\n", a.M, a.N, a.K);
            for (int g = 0; g < 2; ++g)
                for (int st = 0; st < 10; st += 3) {
                    fprintf(stderr, "  g%d s%2d:", g, st + 8);
                    for (int k = 0; k < 6; ++k) fprintf(stderr, " %7u", hb[g * 64 + st * 6 + k] - hb[0]);
                    fprintf(stderr, "\n");
                }
            fprintf(stderr, "  workgroup 0 epilogue (10 ns units after loop end): ring free %u, pass0 staged %u, pass0 stores issued %u, pass1 staged %u, pass1 stores issued %u\n",
                    hb[104], hb[105], hb[106], hb[107], hb[108]);
            fprintf(stderr, "  workgroup 0 main loop: %u s_memtime counts in %.2f us -> %.0f MHz\n", hb[109], hb[110] * 0.01, hb[109] / (hb[110] * 0.01));
            // per-workgroup wall-clock stamps (100 MHz): entry, first barrier passed, loop end, stores acknowledged
            const unsigned long long* r = (const unsigned long long*)(hbuf.data() + 512);
            unsigned long long t_min = ~0ull, t_max = 0;
            for (int i = 0; i < grid; ++i) { t_min = std::min(t_min, r[i * 5]); t_max = std::max(t_max, r[i * 5 + 3]); }
            double pro = 0, loop = 0, epi = 0;
            for (int i = 0; i < grid; ++i) {
                pro += (double)(r[i * 5 + 1] - r[i * 5]); loop += (double)(r[i * 5 + 2] - r[i * 5 + 1]); epi += (double)(r[i * 5 + 3] - r[i * 5 + 2]);
            }
            fprintf(stderr, "  span %.2f us; per workgroup avg: prologue %.2f us, loop %.2f us, epilogue %.2f us\n", (t_max - t_min) * 0.01,
                    pro / grid * 0.01, loop / grid * 0.01, epi / grid * 0.01);
            // timeline of the workgroups that ran on the CU of workgroup 0 (same XCC + HW_ID CU/SE bits)
            auto cu_key = [&](int i) { const unsigned long long v = r[i * 5 + 4]; return (v >> 32 << 16) | ((v >> 8) & 0xff) | (((v >> 13) & 7) << 8); };
            for (int probe : {0, 1}) {
                fprintf(stderr, "  workgroups sharing the CU of workgroup %d (entry, barrier0, loop end, done; us from first entry):\n", probe);
                std::vector<int> ids;
                for (int i = 0; i < grid; ++i) if (cu_key(i) == cu_key(probe)) ids.push_back(i);
                std::sort(ids.begin(), ids.end(), [&](int x, int y) { return r[x * 5] < r[y * 5]; });
                for (int i : ids)
                    fprintf(stderr, "    wg %4d: %7.2f %7.2f %7.2f %7.2f\n", i, (r[i * 5] - t_min) * 0.01, (r[i * 5 + 1] - t_min) * 0.01,
                            (r[i * 5 + 2] - t_min) * 0.01, (r[i * 5 + 3] - t_min) * 0.01);
            }
            // distribution of entry times
            std::vector<double> ent(grid), fin(grid);
            for (int i = 0; i < grid; ++i) { ent[i] = (r[i * 5] - t_min) * 0.01; fin[i] = (r[i * 5 + 3] - t_min) * 0.01; }
            std::sort(ent.begin(), ent.end()); std::sort(fin.begin(), fin.end());
            fprintf(stderr, "  entry times: min %.2f p25 %.2f p50 %.2f p75 %.2f max %.2f; done: min %.2f p50 %.2f max %.2f\n", ent[0], ent[grid / 4], ent[grid / 2],
                    ent[3 * grid / 4], ent[grid - 1], fin[0], fin[grid / 2], fin[grid - 1]);
            return 0;
        }
    }
#endif
    hipLaunchKernelGGL((gemm16v5_kernel<T16, BN, NSLOT>), dim3(grid), dim3(512), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// fraction of the last round of workgroup slots that is filled
static double round_eff(long tiles, long slots) { return (double)tiles / (double)(((tiles + slots - 1) / slots) * slots); }

#define WM_BY_PREC(call_bf16, call_fp16) (prec == WM_PREC_FP16 ? (call_fp16) : (call_bf16))

// The staggered 256 x 320 / 256 x 256 kernel (gemm16_v5.h) serves the shape: the only kernel that reads operands in
// LDS-image order and writes its 16-bit output in it.  Few tiles (one or two image tiles per call): the half-width
// 256 x 160 / 128 kernel fills more CUs; cost model from tools/gemm_bench.py --batch 1: rounds of 256 workgroups x (1.0 | 0.6) per tile.
static bool gemm16_takes_v5(int M, int N, int K) {
    if (M <= 0 || M % 256 || K % G16_BK || K / 32 < 2) return false;
    auto prefer_half = [&](int bn) {
        const long t = (long)(M / 256) * (N / bn);
        return (double)((2 * t + 255) / 256) * 0.6 < (double)((t + 255) / 256);
    };
    if (N % 320 == 0) return !prefer_half(320);
    if (N % 256 == 0) return !prefer_half(256);
    return false;
}

// Options of a launch that only the gemm16_v5 kernel has (the caller asks gemm16_takes_v5 first; a mismatch is an error):
//   Wp: the weight in LDS-image order (or null); a_packed / out_packed: A is / the 16-bit output shall be in that order;
//   st_stats: folded-LayerNorm PRODUCER (fp32 + residual form): per-row partial statistics out, out16 = 16-bit copy of the
//             rows in LDS-image order;
//   fold_stats / fold_c1 / fold_eps: folded-LayerNorm CONSUMER (16-bit-only form): A = such a copy, W = gamma (.) W, bias = c2.
//   res_hi / res_lo / out_lo: split stream (gemm16_v5.h "Split stream"): with st_stats, the residual as two 16-bit planes in
//             (res_hi, res_lo; no fp32 residual) and out (out16 = hi, out_lo); out_lo alone: the fp32-residual producer also
//             writes the lo plane (and no fp32 output when out32 is null).
struct GemmExtra {
    const void* Wp = nullptr;
    int a_packed = 0, out_packed = 0;
    float* st_stats = nullptr;
    const float* fold_stats = nullptr;
    const float* fold_c1 = nullptr;
    float fold_eps = 0.f;
    const void* res_hi = nullptr;
    const void* res_lo = nullptr;
    void* out_lo = nullptr;
    int* overflow = nullptr;
};
static GemmExtra GX(const void* Wp, int a_packed = 0, int out_packed = 0) {
    GemmExtra x;
    x.Wp = Wp; x.a_packed = a_packed; x.out_packed = out_packed;
    return x;
}
// column-tile width of the folded LayerNorm's partial statistics over C channels (the producer GEMM's tile width at N = C)
static int fold_bn_for(int C) { return C % 320 == 0 ? 320 : 256; }

template <class T16, int BN, bool SPLIT = false>
int launch_gemm16v5_foldp_t(wm_handle* h, hipStream_t s, const Gemm16Args& a_in) {
    Gemm16Args a = a_in;
    dev_group_m(a);
    using G = G3<BN, 4>;
    constexpr int LDS = 3 * G::STAGE + 32 * BN * 4;
    WM_TRY(set_max_lds((const void*)gemm16v5_kernel<T16, BN, 3, false, true, false, SPLIT>, LDS));
    count_variant(SPLIT ? (BN == 320 ? WM_GEMM_V5_320_SPLIT : WM_GEMM_V5_256_SPLIT) : (BN == 320 ? WM_GEMM_V5_320_FOLDP : WM_GEMM_V5_256_FOLDP));
    const int grid = (a.M / 256) * (a.N / BN);
    // algorithmic bytes: operands + the stream in and out (split: 2 + 2 B in, 2 + 2 B out; fp32: 4 in, 4 + 2 out, or 2 + 2 out with a lo plane)
    const double stream_bytes = SPLIT ? 8.0 : 4.0 + (a.out32 ? 4.0 : 0.0) + 2.0 + (a.out_lo ? 2.0 : 0.0);
    Bracket br(h, s, WM_KCLASS_GEMM16, 2.0 * a.M * (double)a.N * a.K,
               2.0 * ((double)a.M * a.K + (double)a.N * a.K) + stream_bytes * a.M * a.N);
    hipLaunchKernelGGL((gemm16v5_kernel<T16, BN, 3, false, true, false, SPLIT>), dim3(grid), dim3(512), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T16, int BN>
int launch_gemm16v5_foldc_t(wm_handle* h, hipStream_t s, const Gemm16Args& a_in) {
    Gemm16Args a = a_in;
    dev_group_m(a);
    using G = G3<BN, 4>;
    constexpr int LDS = 3 * G::STAGE + 32 * BN * 4;
    WM_TRY(set_max_lds((const void*)gemm16v5_kernel<T16, BN, 3, false, false, true>, LDS));
    count_variant(BN == 320 ? WM_GEMM_V5_320 : WM_GEMM_V5_256);       // the 16-bit-output instance, with the folded LayerNorm's epilogue
    const int grid = (a.M / 256) * (a.N / BN);
    Bracket br(h, s, WM_KCLASS_GEMM16, 2.0 * a.M * (double)a.N * a.K, 2.0 * ((double)a.M * a.K + (double)a.N * a.K) + 2.0 * a.M * a.N);
    hipLaunchKernelGGL((gemm16v5_kernel<T16, BN, 3, false, false, true>), dim3(grid), dim3(512), LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gemm16(wm_handle* h, hipStream_t s, int prec, const void* A, const void* W, const float* bias,
                  const float* res, int res_mod, float* out32, void* out16, int M, int N, int K, int act,
                  const GemmExtra& x = GemmExtra{}) {
    const void* Wp = x.Wp;
    const int a_packed = x.a_packed, out_packed = x.out_packed;
    if (M <= 0 || N <= 0 || K <= 0 || M % G16_BM || N % G16_BN || K % G16_BK)
        return fail("gemm16: shape M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, G16_BM, G16_BN, G16_BK);
    if (!out32 && !out16) return fail("gemm16: no output");
    Gemm16Args a{};
    a.A = (const u16*)A; a.W = (const u16*)W; a.bias = bias; a.residual = res; a.out32 = out32; a.out16 = (u16*)out16;
    a.M = M; a.N = N; a.K = K; a.res_mod = res_mod; a.act = act;
    if (gemm16_takes_v5(M, N, K)) {         // staggered wave groups (gemm16_v5.h)
        if (Wp) { a.W = (const u16*)Wp; a.w_packed = 1; }
        a.a_packed = a_packed;
        a.out_packed = out_packed;
        if (out_packed && (out32 || res || !out16)) return fail("gemm16: a packed output is the 16-bit-only form (no fp32 output, no residual)");
        if (x.st_stats) {                   // folded LayerNorm, producer
            const bool split = x.res_hi != nullptr;
            if (!out16 || act != ACT_NONE || N / (N % 320 == 0 ? 320 : 256) > 4 || N % 32)
                return fail("gemm16: the statistics-producing form has a 16-bit copy, no activation, at most 4 column tiles");
            if (split ? (res || out32 || !x.res_lo || !x.out_lo || res_mod) : (!res || (!out32 && !x.out_lo)))
                return fail("gemm16: the statistics-producing form takes an fp32 residual (fp32 and / or lo-plane output) or the two planes of a split stream (planes out)");
            a.st_stats = x.st_stats;
            a.res_hi = (const u16*)x.res_hi; a.res_lo = (const u16*)x.res_lo; a.out_lo = (u16*)x.out_lo; a.overflow = x.overflow;
            if (split) {
                if (N % 320 == 0) return WM_BY_PREC((launch_gemm16v5_foldp_t<BF16, 320, true>(h, s, a)), (launch_gemm16v5_foldp_t<FP16, 320, true>(h, s, a)));
                return WM_BY_PREC((launch_gemm16v5_foldp_t<BF16, 256, true>(h, s, a)), (launch_gemm16v5_foldp_t<FP16, 256, true>(h, s, a)));
            }
            if (N % 320 == 0) return WM_BY_PREC((launch_gemm16v5_foldp_t<BF16, 320>(h, s, a)), (launch_gemm16v5_foldp_t<FP16, 320>(h, s, a)));
            return WM_BY_PREC((launch_gemm16v5_foldp_t<BF16, 256>(h, s, a)), (launch_gemm16v5_foldp_t<FP16, 256>(h, s, a)));
        }
        if (x.fold_stats) {                 // folded LayerNorm, consumer
            const int bn = fold_bn_for(K);
            if (out32 || res || !out16 || !x.fold_c1 || !bias || (act != ACT_NONE && act != ACT_GELU) || K % bn || K / bn > 4)
                return fail("gemm16: the folded-LayerNorm form is 16-bit-only output, act none | GELU, K a multiple of %d with at most 4 tiles", bn);
            a.fold_stats = x.fold_stats; a.fold_c1 = x.fold_c1; a.fold_ntile = K / bn; a.fold_bn = (float)bn; a.fold_eps = x.fold_eps;
            if (N % 320 == 0) return WM_BY_PREC((launch_gemm16v5_foldc_t<BF16, 320>(h, s, a)), (launch_gemm16v5_foldc_t<FP16, 320>(h, s, a)));
            return WM_BY_PREC((launch_gemm16v5_foldc_t<BF16, 256>(h, s, a)), (launch_gemm16v5_foldc_t<FP16, 256>(h, s, a)));
        }
        if (N % 320 == 0) return WM_BY_PREC((launch_gemm16v5_t<BF16, 320>(h, s, a)), (launch_gemm16v5_t<FP16, 320>(h, s, a)));
        return WM_BY_PREC((launch_gemm16v5_t<BF16, 256>(h, s, a)), (launch_gemm16v5_t<FP16, 256>(h, s, a)));
    }
    if (a_packed || out_packed || x.st_stats || x.fold_stats || x.res_hi || x.out_lo)
        return fail("gemm16: M=%d N=%d K=%d runs on a half-width kernel, which takes row-major operands only", M, N, K);
    if (M % 256 == 0) {
        // half-width tiles: 256 x 160 where N allows and it fills the last round at least as well as 256 x 128
        const bool can160 = N % 160 == 0;
        const bool use160 = can160 && (N % 320 == 0 || round_eff((long)(M / 256) * (N / 160), 256) >= round_eff((long)(M / 256) * (N / 128), 256) - 1e-9);
        if (use160) return WM_BY_PREC((launch_gemm16v2_t<BF16, 160>(h, s, a)), (launch_gemm16v2_t<FP16, 160>(h, s, a)));
        return WM_BY_PREC((launch_gemm16v2_t<BF16, 128>(h, s, a)), (launch_gemm16v2_t<FP16, 128>(h, s, a)));
    }
    return WM_BY_PREC((launch_gemm16_t<BF16>(h, s, a)), (launch_gemm16_t<FP16>(h, s, a)));
}

// fp8 GEMM (gemm8.h).  prec16 = type of a 16-bit output.  K-step 128 (the 64-byte / 4-slot variant measured equal or
// 1-3 % slower: tools/experiments/gemm8_bk64.h, profiles/r2_dev/gemm8_bench_b16_bk64.txt).
template <class T16, int BKB, bool PLANES = false>
int launch_gemm8_t(wm_handle* h, hipStream_t s, Gemm8Args a, int grid, double flops, double bytes) {
    using G = G8<BKB>;
#if WM_DEV_TIMELINE
    static const bool dbg = getenv("WM_GEMM8_DBG") != nullptr;          // dev: per-workgroup wall-clock stamps of the 5th launch
    static int dbg_count = 0;
    if (!PLANES && dbg && ++dbg_count == 5) {
        unsigned long long* buf = nullptr;
        HIP_TRY(hipMalloc((void**)&buf, (size_t)grid * 32 + 16 + 256));
        HIP_TRY(hipMemset(buf, 0, (size_t)grid * 32 + 16 + 256));
        a.dbg = buf;
        WM_TRY(set_max_lds((const void*)gemm8_kernel<T16, BKB, true>, G::LDS));
        hipLaunchKernelGGL((gemm8_kernel<T16, BKB, true>), dim3(grid), dim3(512), G::LDS, s, a);
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> r((size_t)grid * 4 + 2 + 32);
        HIP_TRY(hipMemcpy(r.data(), buf, r.size() * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(buf));
        unsigned long long t_min = ~0ull, t_max = 0;
        double pro = 0, loop = 0, epi = 0;
        for (int i = 0; i < grid; ++i) {
            t_min = std::min(t_min, r[i * 4]); t_max = std::max(t_max, r[i * 4 + 3]);
            pro += (double)(r[i * 4 + 1] - r[i * 4]); loop += (double)(r[i * 4 + 2] - r[i * 4 + 1]); epi += (double)(r[i * 4 + 3] - r[i * 4 + 2]);
        }
        {
            const unsigned* mk = (const unsigned*)(r.data() + (size_t)grid * 4 + 2);
            for (int g = 0; g < 2; ++g)
                for (int st = 0; st < 4; ++st) {
                    fprintf(stderr, "  [gemm8 dbg] group %d step %d marks (cycles rel. to group-0 step-4 mark 0):", g, st + 4);
                    for (int k = 0; k < 6; ++k) fprintf(stderr, " %7d", (int)(mk[g * 24 + st * 6 + k] - mk[0]));
                    fprintf(stderr, "\n");
                }
        }
        fprintf(stderr, "[gemm8 dbg] workgroup 0 main loop: %llu shader cycles in %.2f us -> %.0f MHz\n", r[(size_t)grid * 4], r[(size_t)grid * 4 + 1] * 0.01,
                (double)r[(size_t)grid * 4] / (r[(size_t)grid * 4 + 1] * 0.01));
        fprintf(stderr, "[gemm8 dbg] BK=%d M=%d N=%d K=%d out=%s: span %.2f us; per workgroup avg: prologue %.2f us, loop %.2f us (%.0f ns per 128 of K), epilogue %.2f us; %d workgroups\n",
                BKB, a.M, a.N, a.K, a.residual ? "f32+res" : (a.out8 ? "fp8" : "16"), (t_max - t_min) * 0.01, pro / grid * 0.01, loop / grid * 0.01,
                loop / grid * 10.0 / (a.K / 128.0), epi / grid * 0.01, grid);
        return 0;
    }
#endif
    WM_TRY(set_max_lds((const void*)gemm8_kernel<T16, BKB, false, PLANES>, G::LDS));
    Bracket br(h, s, WM_KCLASS_GEMM16, flops, bytes);
    hipLaunchKernelGGL((gemm8_kernel<T16, BKB, false, PLANES>), dim3(grid), dim3(512), G::LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gemm8(wm_handle* h, hipStream_t s, int prec16, const void* A, const void* W, const float* wscale, const float* bias,
                 const float* res, float* out32, void* out16, void* out8, int M, int N, int K, int act, void* hi = nullptr, void* lo = nullptr) {
    if (M <= 0 || N <= 0 || K <= 0 || M % G8_BM || N % G8_BN || K % 128 || K < 256)
        return fail("gemm8: shape M=%d N=%d K=%d must be multiples of %d/%d/128 with K >= 256", M, N, K, G8_BM, G8_BN);
    if (!A || !W || !wscale) return fail("gemm8: null operand");
    if (hi || lo) {                                         // the stream's row-major planes, updated in place (gemm8.h PLANES)
        if (!hi || !lo || res || out32 || out16 || out8 || act != ACT_NONE) return fail("gemm8: the plane form takes (hi, lo) and nothing else");
        Gemm8Args a{(const unsigned char*)A, (const unsigned char*)W, wscale, bias, nullptr, nullptr, nullptr, nullptr, M, N, K, act, nullptr,
                    (const u16*)hi, (const u16*)lo, (u16*)hi, (u16*)lo};
        count_variant(WM_GEMM_FP8_256_PLANES);
        const double flops = 2.0 * M * (double)N * K, bytes = (double)M * K + (double)N * K + 8.0 * M * N;
        if (prec16 == WM_PREC_FP16) return launch_gemm8_t<FP16, 128, true>(h, s, a, (M / G8_BM) * (N / G8_BN), flops, bytes);
        return launch_gemm8_t<BF16, 128, true>(h, s, a, (M / G8_BM) * (N / G8_BN), flops, bytes);
    }
    const int modes = (res != nullptr) + (out8 != nullptr) + (res == nullptr && out8 == nullptr && out16 != nullptr);
    if (modes != 1 || (res && !out32 && !out16) || (!res && out32)) return fail("gemm8: outputs must be (residual + out32 [+ out16]) | out8 | out16");
    Gemm8Args a{(const unsigned char*)A, (const unsigned char*)W, wscale, bias, res, out32, (u16*)out16, (unsigned char*)out8, M, N, K, act, nullptr,
                nullptr, nullptr, nullptr, nullptr};
    const int grid = (M / G8_BM) * (N / G8_BN);
    count_variant(WM_GEMM_FP8_256);
    const double flops = 2.0 * M * (double)N * K;
    const double bytes = (double)M * K + (double)N * K + (res ? 8.0 : 0.0) * M * N + (out16 ? 2.0 : 0.0) * M * N + (out8 ? 1.0 : 0.0) * M * N;
    if (prec16 == WM_PREC_FP16) return launch_gemm8_t<FP16, 128>(h, s, a, grid, flops, bytes);
    return launch_gemm8_t<BF16, 128>(h, s, a, grid, flops, bytes);
}

// 3x3 / pad 1 convolution over an NHWC [B,64,64,C] 16-bit activation as an implicit GEMM (no im2col buffer):
// out[M = B*4096, N] = conv(A) with W packed [N][tap][C]  (image_encoder.py:113-119)
int launch_conv3x3_16(wm_handle* h, hipStream_t s, int prec, const void* A, const void* W, float* out32, int M, int N, int Cin) {
    if (M % 4096 || N % 256 || Cin % 32) return fail("conv3x3: M=%d N=%d C=%d unsupported (M %% 4096, N %% 256, C %% 32)", M, N, Cin);
    const uint16_t* zero_page = nullptr;       // 256 B of zeros for out-of-image taps (one per device)
    WM_TRY(zero_page_for_device(&zero_page));
    Gemm16Args a{};
    a.A = (const u16*)A; a.W = (const u16*)W; a.out32 = out32; a.M = M; a.N = N; a.K = 9 * Cin; a.act = ACT_NONE; a.conv_c = Cin; a.zero_page = (const u16*)zero_page;
    using G = G3<256, 4>;
    WM_TRY(set_max_lds((const void*)gemm16v3_kernel<FP16, 256, 4, 1>, G::LDS));
    WM_TRY(set_max_lds((const void*)gemm16v3_kernel<BF16, 256, 4, 1>, G::LDS));
    count_variant(WM_GEMM_V3_CONV3X3);
    Bracket br(h, s, WM_KCLASS_GEMM16, 2.0 * M * (double)N * 9 * Cin, 2.0 * ((double)M * Cin + 9.0 * N * Cin) + 4.0 * M * N);
    if (prec == WM_PREC_FP16) hipLaunchKernelGGL((gemm16v3_kernel<FP16, 256, 4, 1>), dim3((M / 256) * (N / 256)), dim3(G::THREADS), G::LDS, s, a);
    else hipLaunchKernelGGL((gemm16v3_kernel<BF16, 256, 4, 1>), dim3((M / 256) * (N / 256)), dim3(G::THREADS), G::LDS, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// 16 x 16 / stride-16 patch embed as an implicit GEMM (gemm16_v3.h AMODE 2): img16 [B][Cin][1024][1024] 16-bit, W [N][Cin * 256]
// row-major, out[M = B * 4096][N] = patches W^T + bias (+ residual[m % res_mod])  (image_encoder.py:386-450)
int launch_patch_embed16(wm_handle* h, hipStream_t s, int prec, const void* img16, const void* W, const float* bias, const float* res, int res_mod,
                         float* out32, void* out16, int B, int N, int Cin) {
    const int M = B * 4096, K = Cin * 256;
    if (B <= 0 || (N % 320 && N % 256) || Cin <= 0) return fail("patch_embed: B=%d N=%d Cin=%d unsupported (N %% 320 or N %% 256)", B, N, Cin);
    if (!out32 && !out16) return fail("patch_embed: no output");
    Gemm16Args a{};
    a.A = (const u16*)img16; a.W = (const u16*)W; a.bias = bias; a.residual = res; a.out32 = out32; a.out16 = (u16*)out16;
    a.M = M; a.N = N; a.K = K; a.res_mod = res_mod; a.act = ACT_NONE; a.conv_c = Cin;
    count_variant(WM_GEMM_V3_PATCH);
    Bracket br(h, s, WM_KCLASS_GEMM16, 2.0 * M * (double)N * K, 2.0 * ((double)M * K + (double)N * K) + (out32 ? 4.0 : 0.0) * M * N + (out16 ? 2.0 : 0.0) * M * N);
#define WM_PE(T16, BN)                                                                                           \
    do {                                                                                                         \
        using G = G3<BN, 4>;                                                                                     \
        WM_TRY(set_max_lds((const void*)gemm16v3_kernel<T16, BN, 4, 2>, G::LDS));                                \
        hipLaunchKernelGGL((gemm16v3_kernel<T16, BN, 4, 2>), dim3((M / 256) * (N / BN)), dim3(G::THREADS), G::LDS, s, a); \
    } while (0)
    if (N % 320 == 0) { if (prec == WM_PREC_FP16) WM_PE(FP16, 320); else WM_PE(BF16, 320); }
    else { if (prec == WM_PREC_FP16) WM_PE(FP16, 256); else WM_PE(BF16, 256); }
#undef WM_PE
    HIP_TRY(hipGetLastError());
    return 0;
}

unsigned grid_for(int64_t n, int per = 256, unsigned cap = 256 * 16);

// fp32 GEMM (the decoder).  mode: 0 = the engine's choice (the fp16-split form on the 16-bit matrix pipe, gemm32.h gemm32x3_kernel,
// where K % 32 == 0, with W pre-split once per weight upload; WM_GEMM32_F32=1 keeps the fp32-MFMA kernel: A/B runs), 1 = the fp32-MFMA
// kernel, 2 = the split form with W split per K-step (op-level entry: no handle to cache planes in)
int launch_gemm32(wm_handle* h, hipStream_t s, const float* A, const float* W, const float* bias, const float* res,
                  float* out, int M, int N, int K, int act, int lda = 0, int mode = 0) {
    if (K % 16) return fail("gemm32: K=%d must be a multiple of 16", K);
    static const bool f32_only = getenv("WM_GEMM32_F32") && atoi(getenv("WM_GEMM32_F32")) != 0;
    if (mode == 2 && K % 32) return fail("gemm32 (split form): K=%d must be a multiple of 32", K);
    if (mode == 2 || (mode == 0 && h && !f32_only && K % 32 == 0)) {
        Gemm32x3Args a{A, W, nullptr, nullptr, bias, res, out, M, N, K, act, lda > 0 ? lda : K, h ? h->overflow + 1 : nullptr};
        if (mode == 0) {                                    // the weight's fp16 planes: made at first use, dropped with the weights
            auto it = h->w32x3.find(W);
            if (it == h->w32x3.end()) {
                uint16_t *hi = nullptr, *lo = nullptr;
                const size_t n = (size_t)N * K;
                if (n % 4) return fail("gemm32: weight of %zu elements", n);
                WM_TRY(dalloc(h, &hi, n * 2)); WM_TRY(dalloc(h, &lo, n * 2));
                hipLaunchKernelGGL(split_w32_kernel, dim3(grid_for((int64_t)n / 4)), dim3(256), 0, s, W, (u16*)hi, (u16*)lo, (int64_t)n / 4, h->overflow + 1);
                HIP_TRY(hipGetLastError());
                it = h->w32x3.emplace(W, std::make_pair(hi, lo)).first;
            }
            a.Whi = (const u16*)it->second.first; a.Wlo = (const u16*)it->second.second;
        }
        Bracket br(h, s, WM_KCLASS_OTHER, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)M * N) + (mode == 0 ? 4.0 : 4.0) * (double)N * K);
        const dim3 grid(((N + 63) / 64) * ((M + 63) / 64));
        if (mode == 0) hipLaunchKernelGGL(gemm32x3_kernel<true>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(gemm32x3_kernel<false>, grid, dim3(256), 0, s, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    Gemm32Args a{A, W, bias, res, out, M, N, K, act, lda > 0 ? lda : K};
    Bracket br(h, s, WM_KCLASS_OTHER, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    hipLaunchKernelGGL(gemm32_kernel, dim3(((N + 63) / 64) * ((M + 63) / 64)), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_layernorm(wm_handle* h, hipStream_t s, int prec, const float* x, const float* g, const float* b, float eps,
                     float* out32, void* out16, int64_t rows, int C, int nchw_hw = 0) {
    if (C % 256 || C > 1280) return fail("layernorm: C=%d unsupported (multiple of 256, <= 1280)", C);
    if (prec != WM_PREC_BF16 && prec != WM_PREC_FP16)
        return fail("layernorm: precision %d has no fp32-output / general form (e4m3 output exists only for the blocks' 16-bit-only form)", prec);
    const dim3 grid((unsigned)((rows + 3) / 4));
    Bracket br(h, s, WM_KCLASS_LAYERNORM, 0.0, (double)rows * C * (4.0 + (out32 ? 4.0 : 0.0) + (out16 ? 2.0 : 0.0)));
#define LN_CASE(NV)                                                                                                   \
    case NV:                                                                                                          \
        if (prec == WM_PREC_FP16)                                                                                     \
            hipLaunchKernelGGL((layernorm_kernel<FP16, NV>), grid, dim3(256), 0, s, x, g, b, eps, out32, (u16*)out16, rows, nchw_hw); \
        else                                                                                                          \
            hipLaunchKernelGGL((layernorm_kernel<BF16, NV>), grid, dim3(256), 0, s, x, g, b, eps, out32, (u16*)out16, rows, nchw_hw); \
        break;
    switch (C / 256) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5)
        default: return fail("layernorm: C=%d", C);
    }
#undef LN_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}

// LayerNorm of the transformer blocks (norm1 / norm2, 16-bit output): column-tiled statistics, bit-identical to the
// (the same statistics arithmetic as the folded LayerNorm's producers: ln_partial16 / ln_combine).
int launch_layernorm_block(wm_handle* h, hipStream_t s, int prec, const float* x, const float* g, const float* b, float eps,
                           void* out16, int64_t rows, int C, int packed = 0) {
    const int bn = C % 320 == 0 ? 320 : (C % 256 == 0 ? 256 : 0);
    if ((!bn || C / bn > 4) && prec == WM_PREC_FP8) return fail("layernorm: C=%d has no e4m3 form", C);
    if (packed && (prec == WM_PREC_FP8 || !bn || C / bn > 4 || rows % 16 || C % 32))
        return fail("layernorm: no LDS-image-order output for rows=%lld C=%d precision %d", (long long)rows, C, prec);
    if (!bn || C / bn > 4) return launch_layernorm(h, s, prec, x, g, b, eps, nullptr, out16, rows, C);
    const dim3 grid((unsigned)((rows + 3) / 4));
    Bracket br(h, s, WM_KCLASS_LAYERNORM, 0.0, (double)rows * C * (prec == WM_PREC_FP8 ? 5.0 : 6.0));
    if (prec == WM_PREC_FP8) {
        if (bn == 320) hipLaunchKernelGGL((layernorm_tiled_kernel<FP8, 320>), grid, dim3(256), 0, s, x, g, b, eps, (u16*)out16, rows, C, 0);
        else hipLaunchKernelGGL((layernorm_tiled_kernel<FP8, 256>), grid, dim3(256), 0, s, x, g, b, eps, (u16*)out16, rows, C, 0);
    } else if (bn == 320) {
        if (prec == WM_PREC_FP16) hipLaunchKernelGGL((layernorm_tiled_kernel<FP16, 320>), grid, dim3(256), 0, s, x, g, b, eps, (u16*)out16, rows, C, packed);
        else hipLaunchKernelGGL((layernorm_tiled_kernel<BF16, 320>), grid, dim3(256), 0, s, x, g, b, eps, (u16*)out16, rows, C, packed);
    } else {
        if (prec == WM_PREC_FP16) hipLaunchKernelGGL((layernorm_tiled_kernel<FP16, 256>), grid, dim3(256), 0, s, x, g, b, eps, (u16*)out16, rows, C, packed);
        else hipLaunchKernelGGL((layernorm_tiled_kernel<BF16, 256>), grid, dim3(256), 0, s, x, g, b, eps, (u16*)out16, rows, C, packed);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// LayerNorm of the fp8 blocks on the stream's hi plane: plane order in, e4m3 in plane order out (layernorm_plane_fp8_kernel)
int launch_layernorm_plane8(wm_handle* h, hipStream_t s, int in16, const void* hi, const float* g, const float* b, float eps, void* out8, int64_t rows, int C) {
    if (C % 256 || C > 1536 || C < 512 || (in16 != WM_PREC_BF16 && in16 != WM_PREC_FP16)) return fail("layernorm (plane): C=%d type %d", C, in16);
    const dim3 grid((unsigned)((rows + 3) / 4));
    Bracket br(h, s, WM_KCLASS_LAYERNORM, 0.0, (double)rows * C * 3.0);
#define WM_LNP(TIN, NJ) hipLaunchKernelGGL((layernorm_plane_fp8_kernel<TIN, NJ, 1>), grid, dim3(256), 0, s, (const u16*)hi, g, b, eps, (unsigned char*)out8, rows, C)
    if (C > 1024) { if (in16 == WM_PREC_FP16) WM_LNP(FP16, 3); else WM_LNP(BF16, 3); }
    else { if (in16 == WM_PREC_FP16) WM_LNP(FP16, 2); else WM_LNP(BF16, 2); }
#undef WM_LNP
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T16, int HD, bool REL>
int launch_attn_global_t(wm_handle* h, hipStream_t s, const AttnArgs& a, int batch, int kclass) {
    // the 8-wave anti-phase kernel (attn_glob8.h); WM_ATTN_4WAVE=1 (read once per process; A/B runs) keeps every shape on the 4-wave one
    static const bool four_wave = getenv("WM_ATTN_4WAVE") && atoi(getenv("WM_ATTN_4WAVE")) != 0;
    // (head_dim 128, the HFC cross-attention: on the 8-wave kernel since round 4 -- with -m through the bias k-step its phases balance,
    // 1086 vs 1257 us on the 4-wave kernel, profiles/r4_dev/attn_kernels_log2_domain.txt)
    if (a.nq % 256 == 0 && a.nk >= 128 && !four_wave) {
        using L8 = Global8Lds<HD, REL>;
        WM_TRY(set_max_lds((const void*)attn_global8_kernel<T16, HD, REL>, L8::TOTAL + (WM_DEV_TIMELINE ? 4096 : 0)));
        Bracket br(h, s, kclass, 4.0 * batch * a.heads * (double)a.nq * a.nk * HD, 0.0);
#if WM_DEV_TIMELINE
        static const bool dbg = getenv("WM_ATTN_DBG") != nullptr;          // dev: phase stamps of workgroup 0 on the 5th launch
        static int dbg_count = 0;
        if (dbg && ++dbg_count == 5) {
            unsigned long long* buf = nullptr;
            HIP_TRY(hipMalloc((void**)&buf, 8 * 64 * 8));
            HIP_TRY(hipMemset(buf, 0, 8 * 64 * 8));
            AttnArgs a2 = a; a2.tl = buf;
            hipLaunchKernelGGL((attn_global8_kernel<T16, HD, REL>), dim3((a.nq / 256) * a.heads * batch), dim3(512), L8::TOTAL + 4096, s, a2);
            HIP_TRY(hipStreamSynchronize(s));
            unsigned long long host[8 * 64];
            HIP_TRY(hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost));
            const unsigned long long t0 = host[0];
            for (int w = 0; w < 8; ++w) {
                fprintf(stderr, "g8 wave %d:", w);
                for (int i = 0; i < 64; ++i) fprintf(stderr, " %lld", (long long)(host[w * 64 + i] - t0));
                fprintf(stderr, "\n");
            }
            hipFree(buf);
            return 0;
        }
        AttnArgs a1 = a; a1.tl = nullptr;
        hipLaunchKernelGGL((attn_global8_kernel<T16, HD, REL>), dim3((a.nq / 256) * a.heads * batch), dim3(512), L8::TOTAL + 4096, s, a1);
#else
        hipLaunchKernelGGL((attn_global8_kernel<T16, HD, REL>), dim3((a.nq / 256) * a.heads * batch), dim3(512), L8::TOTAL, s, a);
#endif
        HIP_TRY(hipGetLastError());
        return 0;
    }
    using L = GlobalLds<HD, REL>;
    WM_TRY(set_max_lds((const void*)attn_global_kernel<T16, HD, REL>, L::TOTAL));
    Bracket br(h, s, kclass, 4.0 * batch * a.heads * (double)a.nq * a.nk * HD, 0.0);
    hipLaunchKernelGGL((attn_global_kernel<T16, HD, REL>), dim3((a.nq / 128) * a.heads * batch), dim3(256), L::TOTAL, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T16>
int launch_attn_global_p(wm_handle* h, hipStream_t s, const AttnArgs& a, int batch, int hd, bool rel, int kclass) {
    if (a.nq % 128 || a.nk % 64) return fail("attention: nq=%d nk=%d must be multiples of 128/64", a.nq, a.nk);
    if (rel && (a.nq != T || a.nk != T)) return fail("attention: rel-pos path needs 4096 queries and keys");
    if (hd == 80 && rel) return launch_attn_global_t<T16, 80, true>(h, s, a, batch, kclass);
    if (hd == 64 && rel) return launch_attn_global_t<T16, 64, true>(h, s, a, batch, kclass);
    if (hd == 128 && !rel) return launch_attn_global_t<T16, 128, false>(h, s, a, batch, kclass);
    if (hd == 64 && !rel) return launch_attn_global_t<T16, 64, false>(h, s, a, batch, kclass);
    if (hd == 80 && !rel) return launch_attn_global_t<T16, 80, false>(h, s, a, batch, kclass);
    return fail("attention: head_dim=%d rel=%d not built (64, 80, 128)", hd, (int)rel);
}

template <class T16, int HD>
int launch_attn_window_t(wm_handle* h, hipStream_t s, const AttnArgs& a, int batch) {
    const int num_cu = num_cus();
    const int nitems = 25 * a.heads * batch;
    const int grid = nitems < num_cu ? nitems : num_cu;
    // (round 4: an 8-wave anti-phase form of this kernel -- key tiles of 64 slots in a 4-slot ring filled by LDS-DMA, SIMD partners one
    // phase apart, ten barriers per item -- was built, is correct and measured 388 vs 274 us per launch: tools/experiments/
    // attn_win8_antiphase_window.h)
    using L = WindowLds<HD>;
    WM_TRY(set_max_lds((const void*)attn_window_kernel<T16, HD>, L::TOTAL + (WM_DEV_TIMELINE ? 4096 : 0)));
    Bracket br(h, s, WM_KCLASS_ATTN_WIN, 4.0 * batch * a.heads * 4096.0 * 196.0 * HD, 0.0);   // useful work only
#if WM_DEV_TIMELINE
    static const bool dbg = getenv("WM_ATTN_DBG") != nullptr;          // dev: phase stamps of workgroup 0 on the 5th launch
    static int dbg_count = 0;
    if (dbg && ++dbg_count == 5) {
        unsigned long long* buf = nullptr;
        HIP_TRY(hipMalloc((void**)&buf, 8 * 64 * 8));
        HIP_TRY(hipMemset(buf, 0, 8 * 64 * 8));
        AttnArgs a2 = a; a2.tl = buf;
        hipLaunchKernelGGL((attn_window_kernel<T16, HD>), dim3(grid), dim3(448), L::TOTAL + 4096, s, a2, nitems);
        HIP_TRY(hipStreamSynchronize(s));
        unsigned long long host[8 * 64];
        HIP_TRY(hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost));
        for (int w = 0; w < 7; ++w) {
            fprintf(stderr, "win wave %d:", w);
            for (int i = 0; i < 48; ++i) fprintf(stderr, " %lld", (long long)(host[w * 64 + i] - host[0]));
            fprintf(stderr, "\n");
        }
        hipFree(buf);
        return 0;
    }
    hipLaunchKernelGGL((attn_window_kernel<T16, HD>), dim3(grid), dim3(448), L::TOTAL + 4096, s, a, nitems);
#else
    hipLaunchKernelGGL((attn_window_kernel<T16, HD>), dim3(grid), dim3(448), L::TOTAL, s, a, nitems);
#endif
    HIP_TRY(hipGetLastError());
    return 0;
}


// The attention kernels take q in the log2 domain, c1 q with c1 = head_dim^-0.5 * log2 e (attn16.h "Scores").  A caller that holds the
// reference's plain q (the single-op entry points) gets a scaled copy in a scratch buffer: a.q / a.q_stride are redirected to it.
int scale_q_copy(hipStream_t s, int prec, AttnArgs& a, int batch, int cols) {
    const int64_t rows = (int64_t)batch * a.nq;
    void* pb = nullptr;
    WM_TRY(op_scratch(s, 2, (size_t)rows * cols * 2, &pb));
    const float c1 = a.scale * 1.44269504088896340736f;
    const dim3 grid(grid_for(rows * (cols / 8)));
    if (prec == WM_PREC_FP16) hipLaunchKernelGGL(scale_q16_kernel<FP16>, grid, dim3(256), 0, s, a.q, a.q_stride, (u16*)pb, rows, cols, c1);
    else hipLaunchKernelGGL(scale_q16_kernel<BF16>, grid, dim3(256), 0, s, a.q, a.q_stride, (u16*)pb, rows, cols, c1);
    HIP_TRY(hipGetLastError());
    a.q = (const u16*)pb;
    a.q_stride = cols;
    return 0;
}

// q_prescaled: q already carries softmax scale * log2 e (the engine folds it into the q rows of the qkv weight at wm_finalize_weights,
// attn16.h "Scores"); 0 for the single-op entry points, whose callers pass the reference's plain q
int launch_encoder_attention(wm_handle* h, hipStream_t s, int prec, const void* qkv, const float* qkv_bias,
                             const float* rel_h, const float* rel_w, void* out, int batch, int heads, int hd, int window, void* out8 = nullptr,
                             const void* k_sep = nullptr, const void* v_sep = nullptr, int tok_stride = 0, int q_prescaled = 0) {
    const int D = heads * hd;
    AttnArgs a{};
    a.out8 = (unsigned char*)out8;
    a.q = (const u16*)qkv; a.k = (const u16*)qkv + D; a.v = (const u16*)qkv + 2 * D;
    a.out = (u16*)out;
    a.q_stride = a.k_stride = a.v_stride = 3 * D;
    if (k_sep) { a.k = (const u16*)k_sep; a.v = (const u16*)v_sep; a.q_stride = a.k_stride = a.v_stride = tok_stride; }   // q / k / v as three tensors
    a.out_stride = D;
    a.nq = a.nk = T;
    a.scale = 1.0f / sqrtf((float)hd);
    a.rel_h = rel_h; a.rel_w = rel_w; a.qkv_bias = qkv_bias; a.heads = heads;
    if (!q_prescaled) WM_TRY(scale_q_copy(s, prec, a, batch, D));
    if (window == 0) {
        return prec == WM_PREC_FP16 ? launch_attn_global_p<FP16>(h, s, a, batch, hd, true, WM_KCLASS_ATTN_GLOBAL)
                                    : launch_attn_global_p<BF16>(h, s, a, batch, hd, true, WM_KCLASS_ATTN_GLOBAL);
    }
    if (window != 14) return fail("attention: window=%d unsupported (14 or 0)", window);
    {   // the bias as a 16-bit row (AttnArgs::qkv_bias16): cached per handle, converted per call without one
        uint16_t* b16 = nullptr;
        bool convert = true;
        if (h) {
            auto it = h->bias16.find({qkv_bias, prec});
            if (it != h->bias16.end()) { b16 = it->second; convert = false; }
            else { WM_TRY(dalloc(h, &b16, (size_t)3 * D * 2)); h->bias16[{qkv_bias, prec}] = b16; }
        } else {
            void* pb = nullptr;
            WM_TRY(op_scratch(s, 0, (size_t)3 * D * 2, &pb));
            b16 = (uint16_t*)pb;
        }
        if (convert) {
            if (prec == WM_PREC_FP16) hipLaunchKernelGGL(cvt_f32_to_16_kernel<FP16>, dim3((3 * D / 4 + 255) / 256), dim3(256), 0, s, qkv_bias, (u16*)b16, (int64_t)(3 * D / 4));
            else hipLaunchKernelGGL(cvt_f32_to_16_kernel<BF16>, dim3((3 * D / 4 + 255) / 256), dim3(256), 0, s, qkv_bias, (u16*)b16, (int64_t)(3 * D / 4));
            HIP_TRY(hipGetLastError());
        }
        a.qkv_bias16 = (const u16*)b16;
    }
    if (hd == 80) return prec == WM_PREC_FP16 ? launch_attn_window_t<FP16, 80>(h, s, a, batch) : launch_attn_window_t<BF16, 80>(h, s, a, batch);
    if (hd == 64) return prec == WM_PREC_FP16 ? launch_attn_window_t<FP16, 64>(h, s, a, batch) : launch_attn_window_t<BF16, 64>(h, s, a, batch);
    return fail("attention: head_dim=%d not built for windows (64, 80)", hd);
}

int launch_mha16(wm_handle* h, hipStream_t s, int prec, const void* q, int qs, const void* k, int ks, const void* v, int vs,
                 void* out, int os, int batch, int heads, int hd, int nq, int nk, int q_prescaled = 0) {
    AttnArgs a{};
    a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.out = (u16*)out;
    a.q_stride = qs; a.k_stride = ks; a.v_stride = vs; a.out_stride = os;
    a.nq = nq; a.nk = nk; a.scale = 1.0f / sqrtf((float)hd); a.heads = heads;
    if (!q_prescaled) WM_TRY(scale_q_copy(s, prec, a, batch, heads * hd));
    return prec == WM_PREC_FP16 ? launch_attn_global_p<FP16>(h, s, a, batch, hd, false, WM_KCLASS_ATTN_GLOBAL)
                                : launch_attn_global_p<BF16>(h, s, a, batch, hd, false, WM_KCLASS_ATTN_GLOBAL);
}

int launch_mha32(wm_handle* h, hipStream_t s, const float* q, const float* k, const float* v, float* out, int batch,
                 int heads, int hd, int nq, int nk) {
    Bracket br(h, s, WM_KCLASS_OTHER, 4.0 * batch * heads * (double)nq * nk * hd, 0.0);
    // many keys, few queries (token -> image): 4 queries share each K / V row and 4 waves split the keys; otherwise one
    // query per wave
    const bool share = nk >= 1024;
    constexpr int KC = 256;                        // keys per workgroup of the key-split kernel
    if (hd == 16 && nq <= 64 && nk >= 1024 && nk % KC == 0) {
        // token -> image: keys split over workgroups, K / V read once (dec_kernels.h); partials in a scratch buffer of the handle
        // (or, for handle-less op calls, of the process)
        const int nchunk = nk / KC;
        const size_t need = (size_t)batch * heads * nchunk * 64 * (16 + 2) * 4;
        float* part = nullptr;
        if (h) {
            if (h->mha_part_cap < need) {
                if (h->mha_part) { hipFree(h->mha_part); for (auto& a : h->allocs) if (a == h->mha_part) a = nullptr; }     // hipFree synchronises
                h->mha_part = nullptr; h->mha_part_cap = 0;
                WM_TRY(dalloc(h, &h->mha_part, need));
                h->mha_part_cap = need;
            }
            part = h->mha_part;
        } else {
            void* pb = nullptr;
            WM_TRY(op_scratch(s, 1, need, &pb));
            part = (float*)pb;
        }
        hipLaunchKernelGGL((mha32_keysplit_kernel<16, KC>), dim3(nchunk, heads, batch), dim3(256), 0, s, q, k, v, part, nq, nk, heads);
        hipLaunchKernelGGL((mha32_merge_chunks_kernel<16>), dim3(heads, batch), dim3(64 * 4), 0, s, (const float*)part, out, nq, nchunk, heads);
    } else if (hd == 16 && nk == NQ && nq >= 1024)       // image -> token: one thread per query, K / V from scalar loads
        hipLaunchKernelGGL((mha32_fewkeys_kernel<16, NQ>), dim3((nq + 255) / 256, heads, batch), dim3(256), 0, s, q, k, v, out, nq, heads);
    else if (hd == 16 && share) hipLaunchKernelGGL((mha32_kernel<16, 4, 4>), dim3((nq + 3) / 4, heads, batch), dim3(256), 0, s, q, k, v, out, nq, nk, heads);
    else if (hd == 16) hipLaunchKernelGGL((mha32_kernel<16, 1>), dim3(nq, heads, batch), dim3(64), 0, s, q, k, v, out, nq, nk, heads);
    else if (hd == 32) hipLaunchKernelGGL((mha32_kernel<32, 1>), dim3(nq, heads, batch), dim3(64), 0, s, q, k, v, out, nq, nk, heads);
    else return fail("mha32: head_dim=%d not built (16, 32)", hd);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class K, class... Args>
int launch_simple(wm_handle* h, hipStream_t s, double bytes, K kern, dim3 grid, dim3 block, Args... args) {
    Bracket br(h, s, WM_KCLASS_OTHER, 0.0, bytes);
    hipLaunchKernelGGL(kern, grid, block, 0, s, args...);
    HIP_TRY(hipGetLastError());
    return 0;
}

unsigned grid_for(int64_t n, int per, unsigned cap) {
    int64_t g = (n + per - 1) / per;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// -------- expected weights --------
void add_attn(std::map<std::string, std::vector<int64_t>>& m, const std::string& p, int E, int internal) {
    for (const char* n : {"q_proj", "k_proj", "v_proj"}) {
        m[p + n + ".weight"] = {internal, E};
        m[p + n + ".bias"] = {internal};
    }
    m[p + "out_proj.weight"] = {E, internal};
    m[p + "out_proj.bias"] = {E};
}

void build_expected(wm_handle* h) {
    auto& m = h->expected;
    const int D = h->D, hd = h->hd;
    const std::string e = "image_encoder.";
    m[e + "pos_embed"] = {1, GRID, GRID, D};
    m[e + "patch_embed.proj.weight"] = {D, 3, 16, 16};
    m[e + "patch_embed.proj.bias"] = {D};
    m[e + "hfc_embed.proj.weight"] = {HFC, 1, 16, 16};
    m[e + "hfc_embed.proj.bias"] = {HFC};
    const std::string a = e + "hfc_attn.";
    m[a + "pos_embed"] = {1, HFC, GRID, GRID};
    m[a + "proj_hfc.weight"] = {HFC, HFC, 1, 1};
    m[a + "proj_hfc.bias"] = {HFC};
    m[a + "proj_patch.weight"] = {HFC, D, 1, 1};
    m[a + "proj_patch.bias"] = {HFC};
    m[a + "cross_attn.in_proj_weight"] = {3 * HFC, HFC};
    m[a + "cross_attn.in_proj_bias"] = {3 * HFC};
    m[a + "cross_attn.out_proj.weight"] = {HFC, HFC};
    m[a + "cross_attn.out_proj.bias"] = {HFC};
    for (const char* n : {"linear1", "linear2"}) {
        m[a + n + ".weight"] = {HFC, HFC};
        m[a + n + ".bias"] = {HFC};
    }
    for (const char* n : {"norm1", "norm2"}) {
        m[a + n + ".weight"] = {HFC};
        m[a + n + ".bias"] = {HFC};
    }
    m[a + "proj_back.weight"] = {D, HFC, 1, 1};
    m[a + "proj_back.bias"] = {D};
    for (int i = 0; i < h->depth; ++i) {
        const std::string b = e + "blocks." + std::to_string(i) + ".";
        const int size = h->is_global[i] ? GRID : 14;
        m[b + "norm1.weight"] = {D};
        m[b + "norm1.bias"] = {D};
        m[b + "attn.rel_pos_h"] = {2 * size - 1, hd};
        m[b + "attn.rel_pos_w"] = {2 * size - 1, hd};
        m[b + "attn.qkv.weight"] = {3 * D, D};
        m[b + "attn.qkv.bias"] = {3 * D};
        m[b + "attn.proj.weight"] = {D, D};
        m[b + "attn.proj.bias"] = {D};
        m[b + "norm2.weight"] = {D};
        m[b + "norm2.bias"] = {D};
        m[b + "mlp.lin1.weight"] = {4 * D, D};
        m[b + "mlp.lin1.bias"] = {4 * D};
        m[b + "mlp.lin2.weight"] = {D, 4 * D};
        m[b + "mlp.lin2.bias"] = {D};
    }
    m[e + "neck.0.weight"] = {OUTC, D, 1, 1};
    m[e + "neck.1.weight"] = {OUTC};
    m[e + "neck.1.bias"] = {OUTC};
    m[e + "neck.2.weight"] = {OUTC, OUTC, 3, 3};
    m[e + "neck.3.weight"] = {OUTC};
    m[e + "neck.3.bias"] = {OUTC};
    m["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = {2, OUTC / 2};
    const std::string d = "mask_decoder.";
    for (int i = 0; i < 2; ++i) {
        const std::string L = d + "transformer.layers." + std::to_string(i) + ".";
        add_attn(m, L + "self_attn.", OUTC, OUTC);
        add_attn(m, L + "cross_attn_token_to_image.", OUTC, OUTC / 2);
        add_attn(m, L + "cross_attn_image_to_token.", OUTC, OUTC / 2);
        for (const char* n : {"norm1", "norm2", "norm3", "norm4"}) {
            m[L + n + ".weight"] = {OUTC};
            m[L + n + ".bias"] = {OUTC};
        }
        m[L + "mlp.lin1.weight"] = {DEC_MLP, OUTC};
        m[L + "mlp.lin1.bias"] = {DEC_MLP};
        m[L + "mlp.lin2.weight"] = {OUTC, DEC_MLP};
        m[L + "mlp.lin2.bias"] = {OUTC};
    }
    add_attn(m, d + "transformer.final_attn_token_to_image.", OUTC, OUTC / 2);
    m[d + "transformer.norm_final_attn.weight"] = {OUTC};
    m[d + "transformer.norm_final_attn.bias"] = {OUTC};
    m[d + "iou_token.weight"] = {1, OUTC};                 // parameter exists, unused in forward (box_decoder.py:52)
    m[d + "mask_tokens.weight"] = {NQ, OUTC};
    const int cls_dims[4] = {OUTC, OUTC, OUTC, WM_NUM_LOGITS}, box_dims[4] = {OUTC, OUTC, OUTC, 4};
    for (int j = 0; j < 3; ++j) {
        m[d + "class_embed.layers." + std::to_string(j) + ".weight"] = {cls_dims[j + 1], cls_dims[j]};
        m[d + "class_embed.layers." + std::to_string(j) + ".bias"] = {cls_dims[j + 1]};
        m[d + "bbox_embed.layers." + std::to_string(j) + ".weight"] = {box_dims[j + 1], box_dims[j]};
        m[d + "bbox_embed.layers." + std::to_string(j) + ".bias"] = {box_dims[j + 1]};
    }
}

// Stem (patch / HFC embeds), HFC adaptor and neck always run with fp16 operands: they are 2.9 % of the FLOPs,
// their inputs are normalised (|x| of a few units), and in bf16 they alone cost 1e-3 on the logits (DESIGN.md
// "Precision").  The transformer blocks use the handle's precision (bf16 by default).
// precision of transformer block i
static int block_prec(const wm_handle* h, int i) {
    if (h->prec == WM_PREC_FP8) return (i >= h->depth - h->fp8_bf16_tail || i < h->fp8_bf16_head) ? WM_PREC_BF16 : WM_PREC_FP8;
    return (h->prec == WM_PREC_BF16 && i >= h->depth - h->fp16_tail) ? WM_PREC_FP16 : h->prec;
}
static bool is_fp8_block_gemm(const wm_handle* h, const std::string& name) {
    const std::string pre = "image_encoder.blocks.";
    if (h->prec != WM_PREC_FP8 || name.rfind(pre, 0) != 0) return false;
    if (block_prec(h, atoi(name.c_str() + pre.size())) != WM_PREC_FP8) return false;
    const std::pair<const char*, int> sufs[] = {{"attn.qkv.weight", WM_FP8_QKV}, {"attn.proj.weight", WM_FP8_PROJ},
                                                {"mlp.lin1.weight", WM_FP8_MLP}, {"mlp.lin2.weight", WM_FP8_MLP}};
    for (const auto& sf : sufs) {
        const size_t n = strlen(sf.first);
        if (name.size() >= n && name.compare(name.size() - n, n, sf.first) == 0) return (h->fp8_gemms & sf.second) != 0;
    }
    return false;
}
static bool is_fp16_block(const wm_handle* h, const std::string& name) {
    const std::string pre = "image_encoder.blocks.";
    if (name.rfind(pre, 0) != 0) return false;
    return block_prec(h, atoi(name.c_str() + pre.size())) == WM_PREC_FP16;
}

bool is_stem_or_neck(const std::string& name) {
    return name.rfind("image_encoder.patch_embed.", 0) == 0 || name.rfind("image_encoder.hfc_embed.", 0) == 0 ||
           name.rfind("image_encoder.hfc_attn.", 0) == 0 || name.rfind("image_encoder.neck.", 0) == 0;
}

int upload16(wm_handle* h, const std::string& key, const float* src, size_t n) {
    std::vector<uint16_t> tmp(n);
    if (h->prec == WM_PREC_FP16 || is_stem_or_neck(key) || is_fp16_block(h, key)) for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_f16_host(src[i]);
    else for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_bf16_host(src[i]);
    uint16_t* d = nullptr;
    WM_TRY(dalloc(h, &d, n * 2));
    HIP_TRY(hipMemcpy(d, tmp.data(), n * 2, hipMemcpyHostToDevice));
    h->w16[key] = d;
    return 0;
}

// [N][K] fp32 -> e4m3 with one fp32 scale per output channel (absmax / 448), gemm8.h
int upload8(wm_handle* h, const std::string& key, const float* src, size_t rows, size_t cols) {
    std::vector<uint8_t> q(rows * cols);
    std::vector<float> sc(rows);
    for (size_t r = 0; r < rows; ++r) {
        float amax = 0.f;
        for (size_t c = 0; c < cols; ++c) amax = fmaxf(amax, fabsf(src[r * cols + c]));
        const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
        sc[r] = scale;
        for (size_t c = 0; c < cols; ++c) q[r * cols + c] = f32_to_e4m3_host(src[r * cols + c] / scale);
    }
    uint8_t* d = nullptr;
    WM_TRY(dalloc(h, &d, rows * cols));
    HIP_TRY(hipMemcpy(d, q.data(), rows * cols, hipMemcpyHostToDevice));
    h->w8[key] = d;
    const auto ends = [&](const char* suf) { const size_t n = strlen(suf); return key.size() >= n && key.compare(key.size() - n, n, suf) == 0; };
    if (cols % 256 == 0 && (ends("attn.qkv.weight") || ends("mlp.lin1.weight"))) {
        std::vector<uint8_t> qk(rows * cols);
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) qk[r * cols + (size_t)plane_pos((int)c)] = q[r * cols + c];
        uint8_t* dk = nullptr;
        WM_TRY(dalloc(h, &dk, rows * cols));
        HIP_TRY(hipMemcpy(dk, qk.data(), rows * cols, hipMemcpyHostToDevice));
        h->w8k[key] = dk;
    }
    float* ds = nullptr;
    WM_TRY(dalloc(h, &ds, rows * 4));
    HIP_TRY(hipMemcpy(ds, sc.data(), rows * 4, hipMemcpyHostToDevice));
    h->w32[key + ".wscale"] = ds;
    return 0;
}

int upload32(wm_handle* h, const std::string& key, const float* src, size_t n) {
    float* d = nullptr;
    WM_TRY(dalloc(h, &d, n * 4));
    HIP_TRY(hipMemcpy(d, src, n * 4, hipMemcpyHostToDevice));
    h->w32[key] = d;
    return 0;
}

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

}  // namespace

// ---------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------
extern "C" int wm_create(const wm_config* cfg, int device, wm_handle** out) {
    if (!cfg || !out) return fail("wm_create: null argument");
    if (cfg->embed_dim <= 0 || cfg->num_heads <= 0 || cfg->embed_dim % cfg->num_heads) return fail("wm_create: bad dims");
    if (cfg->depth <= 0 || cfg->depth > 64) return fail("wm_create: depth %d out of range", cfg->depth);
    if (cfg->max_batch <= 0) return fail("wm_create: max_batch must be positive");
    if (cfg->embed_dim % 256 || cfg->embed_dim > 1280) return fail("wm_create: embed_dim %d unsupported (multiple of 256, <= 1280)", cfg->embed_dim);
    const int hd = cfg->embed_dim / cfg->num_heads;
    if (hd != 64 && hd != 80) return fail("wm_create: head_dim %d unsupported (64 or 80)", hd);
    if (cfg->precision != WM_PREC_BF16 && cfg->precision != WM_PREC_FP16 && cfg->precision != WM_PREC_FP8) return fail("wm_create: bad precision");
    if (cfg->num_global < 0 || cfg->num_global > WM_MAX_GLOBAL) return fail("wm_create: bad num_global");
    HIP_TRY(hipSetDevice(device));
    wm_handle* h = new wm_handle();
    h->cfg = *cfg; h->device = device;
    // bf16 mode: WM_FP16_TAIL=K gives the last K blocks fp16 operands.  Measured (ViT-H, B=16, one box): K = 0 / 8 / 16 / 32 ->
    // logits 8.2e-4 / 7.9e-4 / 6.5e-4 / 2.4e-4 of the reference at 149.9 / 148.6 / 147.1 / 144.8 tiles/s: every block's bf16
    // rounding contributes alike, so the dial buys margin only in proportion to what it costs; default 0 (= north_star's bf16)
    h->fp16_tail = getenv("WM_FP16_TAIL") ? atoi(getenv("WM_FP16_TAIL")) : 0;
    h->fp8_bf16_tail = getenv("WM_FP8_BF16_TAIL") ? atoi(getenv("WM_FP8_BF16_TAIL")) : 0;
    h->fp8_bf16_head = getenv("WM_FP8_BF16_HEAD") ? atoi(getenv("WM_FP8_BF16_HEAD")) : 0;
    h->row_major = getenv("WM_ROW_MAJOR_OPERANDS") && atoi(getenv("WM_ROW_MAJOR_OPERANDS")) != 0;
    h->fold = (cfg->flags & (WM_CFG_FOLD_LN | WM_CFG_FOLD_LN_BF16)) != 0 && !h->row_major;
    h->fold_bf16 = (cfg->flags & WM_CFG_FOLD_LN_BF16) != 0;
    h->fold_from16 = getenv("WM_FOLD_FROM16") && atoi(getenv("WM_FOLD_FROM16")) != 0;
    h->rows8 = !(getenv("WM_FP8_ROWS") && atoi(getenv("WM_FP8_ROWS")) == 0);
    h->split = h->fold && !(getenv("WM_STREAM_SPLIT") && atoi(getenv("WM_STREAM_SPLIT")) == 0);
    h->fp8_gemms = cfg->fp8_gemms ? (cfg->fp8_gemms & WM_FP8_ALL) : (getenv("WM_FP8_GEMMS") ? (atoi(getenv("WM_FP8_GEMMS")) & WM_FP8_ALL) : WM_FP8_ALL);
    if (cfg->precision == WM_PREC_FP8 && h->fp8_gemms == 0) { delete h; return fail("wm_create: fp8_gemms selects no GEMM"); }
    h->D = cfg->embed_dim; h->depth = cfg->depth; h->heads = cfg->num_heads; h->hd = hd;
    h->prec = cfg->precision; h->maxB = cfg->max_batch;
    for (int i = 0; i < cfg->num_global; ++i) {
        const int g = cfg->global_attn_indexes[i];
        if (g < 0 || g >= cfg->depth) { delete h; return fail("wm_create: global index %d out of range", g); }
        h->is_global[g] = true;
    }
    build_expected(h);

    const size_t B = (size_t)h->maxB, D = (size_t)h->D, BT = B * T;
    int r = 0;
#define A(ptr, bytes) if (!r) r = dalloc(h, &h->ptr, (bytes))
    A(resid, BT * D * 4); A(tokbase, BT * D * 4);
    A(xn16, BT * D * 2); A(ao16, BT * D * 2); A(qkv16, BT * 3 * D * 2); A(hid16, BT * 4 * D * 2);
    A(p16, BT * 768 * 2); A(h16, BT * 256 * 2); A(he16, BT * HFC * 2); A(hp16, BT * HFC * 2); A(pt16, BT * HFC * 2);
    A(q16, BT * HFC * 2); A(kv16, BT * 2 * HFC * 2); A(aoh16, BT * HFC * 2); A(y1n16, BT * HFC * 2); A(h1_16, BT * HFC * 2);
    A(y2_16, BT * HFC * 2); A(y2t16, BT * HFC * 2);
    A(pt32, BT * HFC * 4); A(y1, BT * HFC * 4); A(y1n32, BT * HFC * 4); A(z32, BT * HFC * 4);
    A(n1, BT * OUTC * 4); A(n2, BT * OUTC * 4); A(emb_nhwc, BT * OUTC * 4); A(emb_nchw, BT * OUTC * 4);
    A(n1n16, BT * OUTC * 2); A(x16last, BT * D * 2);
    if (cfg->precision == WM_PREC_FP8) { A(ao8, BT * D); }
    A(dkeys, BT * OUTC * 4); A(dk_a, BT * 128 * 4); A(dk_b, BT * 128 * 4); A(dk_c, BT * 128 * 4);
    A(dq, B * NQ * OUTC * 4); A(dt_q, B * NQ * OUTC * 4); A(dt_k, B * NQ * OUTC * 4); A(dt_v, B * NQ * OUTC * 4);
    A(dt_att, B * NQ * OUTC * 4); A(dt_hid, B * NQ * DEC_MLP * 4); A(dt_h1, B * NQ * OUTC * 4); A(dt_h2, B * NQ * OUTC * 4);
    A(logits, B * NQ * WM_NUM_LOGITS * 4); A(boxes, B * NQ * 4 * 4);
    A(hfc, B * 1024 * 1024 * 4); A(tsz_default, B * 2 * 4);
    A(fftR, B * FFT_N * FFT_L * sizeof(float2)); A(fft_tw, FFT_N * sizeof(float2));
    A(kpe, (size_t)T * OUTC * 4);
    A(records, B * NQ * sizeof(wm_box_record));
    A(sat_counts, WM_SAT_COUNT * sizeof(unsigned long long));
    A(fold_stats, BT * 4 * 2 * 4);
    A(lo16, BT * D * 2);
#undef A
    if (!r) {
        void* pf = nullptr;
        if (hipHostMalloc(&pf, 64, hipHostMallocMapped) != hipSuccess) r = fail("wm_create: hipHostMalloc failed");
        else { h->overflow = (int*)pf; h->overflow[0] = h->overflow[1] = 0; }       // [0] the fp16 stream, [1] the decoder's fp16-split GEMMs
    }
    if (r) { wm_destroy(h); return r; }
    // FFT twiddles exp(-2 pi i k / 1024), computed in double
    {
        std::vector<float2> tw(FFT_N);
        for (int k = 0; k < FFT_N; ++k) {
            const double ang = -2.0 * M_PI * k / FFT_N;
            tw[k] = make_float2((float)cos(ang), (float)sin(ang));
        }
        hipError_t e = hipMemcpy(h->fft_tw, tw.data(), FFT_N * sizeof(float2), hipMemcpyHostToDevice);
        std::vector<float> ts(B * 2, 1024.f);
        if (e == hipSuccess) e = hipMemcpy(h->tsz_default, ts.data(), B * 2 * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(h->sat_counts, 0, WM_SAT_COUNT * sizeof(unsigned long long));
        if (e != hipSuccess) { wm_destroy(h); return fail("wm_create: twiddle upload failed: %s", hipGetErrorString(e)); }
    }
    *out = h;
    return 0;
}

extern "C" int wm_destroy(wm_handle* h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    for (void* p : h->allocs) if (p) hipFree(p);
    if (h->tap_buf) hipFree(h->tap_buf);
    if (h->overflow) hipHostFree(h->overflow);
    for (auto& e : h->prof.used) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto& e : h->prof.pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    delete h;
    return 0;
}

extern "C" int wm_load_weight(wm_handle* h, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    if (!h || !name || !host_data || !shape) return fail("wm_load_weight: null argument");
    auto it = h->expected.find(name);
    if (it == h->expected.end()) return fail("wm_load_weight: unknown tensor '%s'", name);
    const auto& es = it->second;
    bool ok = (int)es.size() == ndim;
    size_t n = 1;
    for (int i = 0; ok && i < ndim; ++i) { ok = es[i] == shape[i]; n *= (size_t)shape[i]; }
    if (!ok) return fail("wm_load_weight: shape mismatch for '%s'", name);
    HostW w;
    w.shape.assign(shape, shape + ndim);
    w.data.assign(host_data, host_data + n);
    h->staged[name] = std::move(w);
    h->finalized = false;
    return 0;
}

extern "C" int wm_finalize_weights(wm_handle* h) {
    if (!h) return fail("wm_finalize_weights: null handle");
    HIP_TRY(hipSetDevice(h->device));
    // Two independently loadable groups: the encoder ("image_encoder.*") and the
    // decoder ("mask_decoder.*" + "prompt_encoder.*").  A group is ready when every
    // tensor of it has been loaded; a partially loaded group is an error.
    for (int grp = 0; grp < 2; ++grp) {
        std::string missing;
        int nmiss = 0, nhave = 0;
        for (auto& kv : h->expected) {
            const bool enc = kv.first.rfind("image_encoder.", 0) == 0;
            if ((grp == 0) != enc) continue;
            if (h->staged.count(kv.first) || h->w16.count(kv.first) || h->w32.count(kv.first) || h->w8.count(kv.first)) { ++nhave; continue; }
            if (nmiss < 4) missing += (nmiss ? ", " : "") + kv.first;
            ++nmiss;
        }
        if (nmiss && nhave)
            return fail("wm_finalize_weights: %s group incomplete, %d tensors missing (%s%s)", grp == 0 ? "encoder" : "decoder",
                        nmiss, missing.c_str(), nmiss > 4 ? ", ..." : "");
        if (grp == 0) h->enc_ready = nmiss == 0;
        else h->dec_ready = nmiss == 0;
    }
    if (!h->enc_ready && !h->dec_ready) return fail("wm_finalize_weights: no weights loaded");
    // Attention scores are computed in the log2 domain with the scale inside q (attn16.h "Scores"): the q rows of every qkv
    // weight and bias (and of the HFC cross-attention's in_proj) are multiplied by head_dim^-0.5 * log2 e here, in fp32, BEFORE the one
    // rounding to the operand type (16-bit, folded gamma (.) W, or e4m3 with its per-channel scale), so q = (c1 q_ref) costs no rounding.
    // Each staged tensor passes here exactly once (the staging area is cleared at the end of this call).
    {
        const float c1_blk = (1.0f / sqrtf((float)h->hd)) * 1.44269504088896340736f;
        const float c1_hfc = (1.0f / sqrtf((float)(HFC / HFC_HEADS))) * 1.44269504088896340736f;
        for (auto& kv : h->staged) {
            const std::string& name = kv.first;
            std::vector<float>& d = kv.second.data;
            float c1 = 0.f;
            size_t nq = 0;                                  // leading elements that belong to q
            if (name.rfind("image_encoder.blocks.", 0) == 0 && ends_with(name, "attn.qkv.weight")) { c1 = c1_blk; nq = (size_t)h->D * h->D; }
            else if (name.rfind("image_encoder.blocks.", 0) == 0 && ends_with(name, "attn.qkv.bias")) { c1 = c1_blk; nq = (size_t)h->D; }
            else if (name == "image_encoder.hfc_attn.cross_attn.in_proj_weight") { c1 = c1_hfc; nq = (size_t)HFC * HFC; }
            else if (name == "image_encoder.hfc_attn.cross_attn.in_proj_bias") { c1 = c1_hfc; nq = (size_t)HFC; }
            for (size_t i = 0; i < nq && i < d.size(); ++i) d[i] *= c1;
        }
    }
    // re-upload: free previous device copies of the tensors being replaced
    for (auto& kv : h->staged) {
        auto i16 = h->w16.find(kv.first);
        if (i16 != h->w16.end()) { hipFree(i16->second); for (auto& a : h->allocs) if (a == i16->second) a = nullptr; h->w16.erase(i16); }
        auto i16p = h->w16p.find(kv.first);
        if (i16p != h->w16p.end()) { hipFree(i16p->second); for (auto& a : h->allocs) if (a == i16p->second) a = nullptr; h->w16p.erase(i16p); }
        auto i32 = h->w32.find(kv.first);
        if (i32 != h->w32.end()) { hipFree(i32->second); for (auto& a : h->allocs) if (a == i32->second) a = nullptr; h->w32.erase(i32); }
        auto is32 = h->wsrc32.find(kv.first);
        if (is32 != h->wsrc32.end()) { hipFree(is32->second); for (auto& a : h->allocs) if (a == is32->second) a = nullptr; h->wsrc32.erase(is32); }
        auto i8 = h->w8.find(kv.first);
        if (i8 != h->w8.end()) { hipFree(i8->second); for (auto& a : h->allocs) if (a == i8->second) a = nullptr; h->w8.erase(i8); }
        auto i8k = h->w8k.find(kv.first);
        if (i8k != h->w8k.end()) { hipFree(i8k->second); for (auto& a : h->allocs) if (a == i8k->second) a = nullptr; h->w8k.erase(i8k); }
        auto isc = h->w32.find(kv.first + ".wscale");
        if (isc != h->w32.end()) { hipFree(isc->second); for (auto& a : h->allocs) if (a == isc->second) a = nullptr; h->w32.erase(isc); }
    }
    // 16-bit copies of the qkv biases are keyed by the fp32 copy's address: a re-upload may reuse an address for new values
    for (auto& kv : h->bias16) { hipFree(kv.second); for (auto& a : h->allocs) if (a == kv.second) a = nullptr; }
    h->bias16.clear();
    for (auto& kv : h->w32x3)
        for (uint16_t* q : {kv.second.first, kv.second.second}) { hipFree(q); for (auto& a : h->allocs) if (a == q) a = nullptr; }
    h->w32x3.clear();
    const int D = h->D;
    for (auto& kv : h->staged) {
        const std::string& name = kv.first;
        const HostW& w = kv.second;
        const size_t n = w.data.size();
        const bool enc = name.rfind("image_encoder.", 0) == 0;
        const bool is_gemm_w = enc && (ends_with(name, ".weight") || ends_with(name, "in_proj_weight")) && w.shape.size() >= 2;
        if (name == "image_encoder.neck.2.weight") {
            // [co][ci][ky][kx] -> [co][tap][ci]
            std::vector<float> t(n);
            for (int co = 0; co < OUTC; ++co)
                for (int ci = 0; ci < OUTC; ++ci)
                    for (int tap = 0; tap < 9; ++tap)
                        t[((size_t)co * 9 + tap) * OUTC + ci] = w.data[((size_t)co * OUTC + ci) * 9 + tap];
            WM_TRY(upload16(h, name, t.data(), n));
        } else if (name == "image_encoder.hfc_attn.pos_embed") {
            // NCHW (1,1024,64,64) -> token-major [4096,1024] (added after proj_hfc, image_encoder.py:494)
            std::vector<float> t(n);
            for (int c = 0; c < HFC; ++c)
                for (int p = 0; p < T; ++p) t[(size_t)p * HFC + c] = w.data[(size_t)c * T + p];
            WM_TRY(upload32(h, name, t.data(), n));
        } else if (is_gemm_w && is_fp8_block_gemm(h, name)) {
            WM_TRY(upload8(h, name, w.data.data(), (size_t)w.shape[0], n / (size_t)w.shape[0]));
        } else if (is_gemm_w) {
            WM_TRY(upload16(h, name, w.data.data(), n));
            // second copy in LDS-image order for the 256-row-tile kernel (row-major stays for the half-width kernels that
            // small batches take): [N][K] with N % 16 == 0 and K % 32 == 0 (every GEMM weight of the encoder)
            const int64_t rows = w.shape[0], cols = (int64_t)n / rows;
            if (rows % 16 == 0 && cols % 32 == 0) {
                uint16_t* dp = nullptr;
                WM_TRY(dalloc(h, &dp, n * 2));
                hipLaunchKernelGGL(pack16_lds_image_kernel, dim3(grid_for((int64_t)n / 8)), dim3(256), 0, 0, (const uint4*)h->w16.at(name), (uint4*)dp, rows, (int)cols);
                HIP_TRY(hipGetLastError());
                h->w16p[name] = dp;
            }
            // a weight the folded LayerNorm multiplies by gamma: keep the fp32 values on the device (1.47 GB for ViT-H, of 288)
            if (h->fold && name.rfind("image_encoder.blocks.", 0) == 0 && (ends_with(name, "attn.qkv.weight") || ends_with(name, "mlp.lin1.weight"))) {
                float* d32 = nullptr;
                WM_TRY(dalloc(h, &d32, n * 4));
                HIP_TRY(hipMemcpy(d32, w.data.data(), n * 4, hipMemcpyHostToDevice));
                h->wsrc32[name] = d32;
            }
        } else {
            WM_TRY(upload32(h, name, w.data.data(), n));
        }
    }
    // dense positional encoding, token-major (pos_encoder.py:50-70)
    if (h->staged.count("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix")) {
        const HostW& g = h->staged["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"];
        const int F = OUTC / 2;
        std::vector<float> pe((size_t)T * OUTC);
        for (int y = 0; y < GRID; ++y)
            for (int x = 0; x < GRID; ++x) {
                const float cx = 2.0f * ((x + 0.5f) / GRID) - 1.0f, cy = 2.0f * ((y + 0.5f) / GRID) - 1.0f;
                for (int f = 0; f < F; ++f) {
                    float arg = cx * g.data[f] + cy * g.data[F + f];
                    arg = arg * 6.283185307179586f;
                    pe[(size_t)(y * GRID + x) * OUTC + f] = (float)sin((double)arg);
                    pe[(size_t)(y * GRID + x) * OUTC + F + f] = (float)cos((double)arg);
                }
            }
        HIP_TRY(hipMemcpy(h->kpe, pe.data(), pe.size() * 4, hipMemcpyHostToDevice));
    }
    // Folded LayerNorm: gamma (.) W (LDS-image order), c1, c2 of every block's qkv (norm1) and lin1 (norm2), from the DEVICE
    // copies of the fp32 weights (kept above), so a later partial re-upload folds to the same bits as a full one.
    if (h->fold && h->enc_ready) {
        for (int i = 0; i < h->depth; ++i) {
            const std::string b = "image_encoder.blocks." + std::to_string(i) + ".";
            const int P = block_prec(h, i) == WM_PREC_FP8 ? WM_PREC_BF16 : block_prec(h, i);
            const std::pair<const char*, const char*> pairs[] = {{"attn.qkv", "norm1"}, {"mlp.lin1", "norm2"}};
            for (const auto& pr : pairs) {
                const std::string wn = b + pr.first + ".weight";
                if (!h->w16.count(wn)) continue;            // an fp8 GEMM of this block: no 16-bit weight, no fold
                const int N = (int)h->expected.at(wn)[0], K = (int)h->expected.at(wn)[1];
                if (!h->wfold.count(wn)) {
                    uint16_t* wf = nullptr; float *c1 = nullptr, *c2 = nullptr;
                    WM_TRY(dalloc(h, &wf, (size_t)N * K * 2)); WM_TRY(dalloc(h, &c1, (size_t)N * 4)); WM_TRY(dalloc(h, &c2, (size_t)N * 4));
                    h->wfold[wn] = wf; h->fold_c1[wn] = c1; h->fold_c2[wn] = c2;
                }
                const float* g = h->w32.at(b + pr.second + ".weight");
                const float* be = h->w32.at(b + pr.second + ".bias");
                const float* bias = h->w32.at(b + pr.first + ".bias");
                if (P == WM_PREC_FP16)
                    hipLaunchKernelGGL(fold_weight_kernel<FP16>, dim3(N), dim3(256), 0, 0, (const u16*)h->w16.at(wn), h->fold_from16 ? (const float*)nullptr : (const float*)h->wsrc32.at(wn), g, be, bias, (u16*)h->wfold[wn], h->fold_c1[wn], h->fold_c2[wn], N, K);
                else
                    hipLaunchKernelGGL(fold_weight_kernel<BF16>, dim3(N), dim3(256), 0, 0, (const u16*)h->w16.at(wn), h->fold_from16 ? (const float*)nullptr : (const float*)h->wsrc32.at(wn), g, be, bias, (u16*)h->wfold[wn], h->fold_c1[wn], h->fold_c2[wn], N, K);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    (void)D;
    HIP_TRY(hipDeviceSynchronize());        // the pack / fold launches above
    h->staged.clear();
    h->finalized = true;
    return 0;
}

// ---------------------------------------------------------------------------
// the path
// ---------------------------------------------------------------------------
namespace {

int check_ready(wm_handle* h, int batch, const char* fn, bool need_enc, bool need_dec) {
    if (!h) return fail("%s: null handle", fn);
    if (!h->finalized) return fail("%s: weights not finalized", fn);
    if (need_enc && !h->enc_ready) return fail("%s: encoder weights not loaded", fn);
    if (need_dec && !h->dec_ready) return fail("%s: decoder weights not loaded", fn);
    if (batch <= 0 || batch > h->maxB) return fail("%s: batch %d outside 1..%d", fn, batch, h->maxB);
    HIP_TRY(hipSetDevice(h->device));
    return 0;
}

const uint16_t* W16(wm_handle* h, const std::string& n) { return h->w16.at(n); }
const uint16_t* W16P(wm_handle* h, const std::string& n) {
    if (h->row_major) return nullptr;
    auto it = h->w16p.find(n);
    return it == h->w16p.end() ? nullptr : it->second;
}
const float* W32(wm_handle* h, const std::string& n) { return h->w32.at(n); }

int do_tap(wm_handle* h, hipStream_t s, int which, int batch, const float* src = nullptr) {
    if (h->tap_which != which) return 0;
    if (!h->tap_buf) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, (size_t)h->maxB * T * h->D * 4));
        h->tap_buf = (float*)p;
    }
    HIP_TRY(hipMemcpyAsync(h->tap_buf, src ? src : h->resid, (size_t)batch * T * h->D * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}
int tap_alloc(wm_handle* h) {
    if (!h->tap_buf) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, (size_t)h->maxB * T * h->D * 4));
        h->tap_buf = (float*)p;
    }
    return 0;
}

int fft_impl(wm_handle* h, const float* x, float* out, int B, hipStream_t s, bool copies16 = false) {
    WM_TRY(launch_simple(h, s, (double)B * (12e6 + 3e6), fft_rows_fwd_kernel, dim3(FFT_N, B), dim3(256), x, h->fftR, (const float2*)h->fft_tw));
    WM_TRY(launch_simple(h, s, (double)B * 6e6, fft_cols_kernel, dim3(FFT_L, B), dim3(256), h->fftR, (const float2*)h->fft_tw));
    // copies16: the last pass also leaves fp16 NCHW copies of x and of the result in p16 / h16 (the patch embeds' operands)
    WM_TRY(launch_simple(h, s, (double)B * (12e6 + 3e6 + 4e6 + (copies16 ? 8e6 : 0.0)), fft_rows_inv_kernel<FP16>, dim3(FFT_N, B), dim3(256), x, (const float2*)h->fftR,
                         (const float2*)h->fft_tw, out, copies16 ? (u16*)h->p16 : (u16*)nullptr, copies16 ? (u16*)h->h16 : (u16*)nullptr));
    return 0;
}

// Folded LayerNorm, standalone producer (ln_stats_x16_kernel): partial statistics + 16-bit copy of `rows` fp32 rows of C channels
int launch_ln_stats16(wm_handle* h, hipStream_t s, int prec, const float* x, float* stats, void* x16, int64_t rows, int C,
                      void* lo16 = nullptr, float* x_rw = nullptr, int* overflow = nullptr) {
    const int bn = fold_bn_for(C);
    if (C % bn || C / bn > 4 || rows % 16 || C % 32 || (prec != WM_PREC_FP16 && prec != WM_PREC_BF16))
        return fail("ln_stats16: rows=%lld C=%d precision %d unsupported", (long long)rows, C, prec);
    const dim3 grid((unsigned)((rows + 3) / 4));
    Bracket br(h, s, WM_KCLASS_LAYERNORM, 0.0, (double)rows * C * (6.0 + (lo16 ? 2.0 : 0.0) + (x_rw ? 4.0 : 0.0)));
    if (bn == 320) {
        if (prec == WM_PREC_FP16) hipLaunchKernelGGL((ln_stats_x16_kernel<FP16, 320>), grid, dim3(256), 0, s, x, stats, (u16*)x16, rows, C, (u16*)lo16, x_rw, overflow);
        else hipLaunchKernelGGL((ln_stats_x16_kernel<BF16, 320>), grid, dim3(256), 0, s, x, stats, (u16*)x16, rows, C, (u16*)lo16, x_rw, overflow);
    } else {
        if (prec == WM_PREC_FP16) hipLaunchKernelGGL((ln_stats_x16_kernel<FP16, 256>), grid, dim3(256), 0, s, x, stats, (u16*)x16, rows, C, (u16*)lo16, x_rw, overflow);
        else hipLaunchKernelGGL((ln_stats_x16_kernel<BF16, 256>), grid, dim3(256), 0, s, x, stats, (u16*)x16, rows, C, (u16*)lo16, x_rw, overflow);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// split stream -> fp32 rows (stream_merge_kernel): out[row][col] = float(hi) + float(lo)
int launch_stream_merge(wm_handle* h, hipStream_t s, int prec, const void* hi, const void* lo, float* out, int64_t rows, int C) {
    if (rows % 16 || C % 32 || (prec != WM_PREC_FP16 && prec != WM_PREC_BF16)) return fail("stream_merge: rows=%lld C=%d precision %d", (long long)rows, C, prec);
    Bracket br(h, s, WM_KCLASS_OTHER, 0.0, (double)rows * C * 8.0);
    const dim3 grid(grid_for(rows * (C / 8)));
    if (prec == WM_PREC_FP16) hipLaunchKernelGGL(stream_merge_kernel<FP16>, grid, dim3(256), 0, s, (const u16*)hi, (const u16*)lo, out, rows, C);
    else hipLaunchKernelGGL(stream_merge_kernel<BF16>, grid, dim3(256), 0, s, (const u16*)hi, (const u16*)lo, out, rows, C);
    HIP_TRY(hipGetLastError());
    return 0;
}

// the fp8 blocks' stream: fp32 rows <-> planes of rows (hi of type prec, lo fp16; column c at wm::plane_pos(c))
int launch_stream_rows(wm_handle* h, hipStream_t s, int prec, float* x32, void* hi, void* lo, int64_t rows, int C, bool merge) {
    if (rows <= 0 || C % 256 || (prec != WM_PREC_FP16 && prec != WM_PREC_BF16)) return fail("stream rows: rows=%lld C=%d precision %d", (long long)rows, C, prec);
    const int64_t n = rows * C;
    Bracket br(h, s, WM_KCLASS_OTHER, 0.0, (double)n * 8.0);
    const dim3 grid(grid_for(n / 4));
    if (merge) {
        if (prec == WM_PREC_FP16) hipLaunchKernelGGL(stream_merge_rows_kernel<FP16>, grid, dim3(256), 0, s, (const u16*)hi, (const u16*)lo, x32, n / 4, C);
        else hipLaunchKernelGGL(stream_merge_rows_kernel<BF16>, grid, dim3(256), 0, s, (const u16*)hi, (const u16*)lo, x32, n / 4, C);
    } else {
        if (prec == WM_PREC_FP16) hipLaunchKernelGGL(stream_split_rows_kernel<FP16>, grid, dim3(256), 0, s, (const float*)x32, (u16*)hi, (u16*)lo, n / 4, C);
        else hipLaunchKernelGGL(stream_split_rows_kernel<BF16>, grid, dim3(256), 0, s, (const float*)x32, (u16*)hi, (u16*)lo, n / 4, C);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// opt-in census of clamped values in a 16-bit / e4m3 activation buffer (wm_debug_saturation_enable); prec = element type
int sat_check(wm_handle* h, hipStream_t s, int which, const void* buf, int64_t n_elems, int prec) {
    if (!h->sat_on) return 0;
    const int64_t bytes = n_elems * (prec == WM_PREC_FP8 ? 1 : 2);
    if (bytes % 16) return fail("saturation census: buffer of %lld bytes", (long long)bytes);
    const unsigned thr = prec == WM_PREC_FP8 ? 0x7eu : (prec == WM_PREC_FP16 ? 0x7bffu : 0x7f7fu);
    if (prec == WM_PREC_FP8)
        hipLaunchKernelGGL(saturation_count_kernel<1>, dim3(grid_for(bytes / 16)), dim3(256), 0, s, (const uint4*)buf, bytes / 16, thr, h->sat_counts + which);
    else
        hipLaunchKernelGGL(saturation_count_kernel<2>, dim3(grid_for(bytes / 16)), dim3(256), 0, s, (const uint4*)buf, bytes / 16, thr, h->sat_counts + which);
    HIP_TRY(hipGetLastError());
    return 0;
}

int encoder_impl(wm_handle* h, const float* x, const float* hfc, float* out_nchw, int B, hipStream_t s, bool have16 = false) {
    const int D = h->D, M = B * T;
    const int PS = WM_PREC_FP16;      // stem, HFC adaptor and neck: fp16 operands in every mode (see is_stem_or_neck)
    const std::string e = "image_encoder.", a = e + "hfc_attn.";
    // ---- stem: patch / HFC embeds (image_encoder.py:124-128) ----
    // the embeds read 16-bit NCHW copies of x and hfc (p16, h16) -- left there by the FFT's last pass (wm_forward) or made here --
    // through the implicit-GEMM loader (gemm16_v3.h AMODE 2): no im2col buffer
    if (!have16) {
        WM_TRY(launch_simple(h, s, B * 18.9e6, cvt_f32_to_16_kernel<FP16>, dim3(grid_for((int64_t)B * 3 * 1024 * 256)), dim3(256), x, (u16*)h->p16, (int64_t)B * 3 * 1024 * 256));
        WM_TRY(launch_simple(h, s, B * 6.3e6, cvt_f32_to_16_kernel<FP16>, dim3(grid_for((int64_t)B * 1024 * 256)), dim3(256), hfc, (u16*)h->h16, (int64_t)B * 1024 * 256));
    }
    // t = patch_embed(x) + pos_embed  -> tokbase (fp32) and xn16 (16-bit copy for proj_patch)
    WM_TRY(launch_patch_embed16(h, s, PS, h->p16, W16(h, e + "patch_embed.proj.weight"), W32(h, e + "patch_embed.proj.bias"),
                                W32(h, e + "pos_embed"), T, h->tokbase, h->xn16, B, D, 3));
    WM_TRY(do_tap(h, s, -3, B, h->tokbase));
    WM_TRY(launch_patch_embed16(h, s, PS, h->h16, W16(h, e + "hfc_embed.proj.weight"), W32(h, e + "hfc_embed.proj.bias"),
                                nullptr, 0, nullptr, h->he16, B, HFC, 1));
    // ---- HFC adaptor (image_encoder.py:486-516) ----
    WM_TRY(launch_gemm16(h, s, PS, h->he16, W16(h, a + "proj_hfc.weight"), W32(h, a + "proj_hfc.bias"),
                         W32(h, a + "pos_embed"), T, nullptr, h->hp16, M, HFC, HFC, ACT_NONE, GX(W16P(h, a + "proj_hfc.weight"))));                    // :494
    WM_TRY(launch_gemm16(h, s, PS, h->xn16, W16(h, a + "proj_patch.weight"), W32(h, a + "proj_patch.bias"),
                         nullptr, 0, h->pt32, h->pt16, M, HFC, D, ACT_NONE, GX(W16P(h, a + "proj_patch.weight"))));                                       // :495
    const uint16_t* wi = W16(h, a + "cross_attn.in_proj_weight");
    const uint16_t* wip = W16P(h, a + "cross_attn.in_proj_weight");      // same element offsets: 16 rows x K are one contiguous block in both layouts
    const float* bi = W32(h, a + "cross_attn.in_proj_bias");
    WM_TRY(launch_gemm16(h, s, PS, h->pt16, wi, bi, nullptr, 0, nullptr, h->q16, M, HFC, HFC, ACT_NONE, GX(wip)));
    WM_TRY(launch_gemm16(h, s, PS, h->hp16, wi + (size_t)HFC * HFC, bi + HFC, nullptr, 0, nullptr, h->kv16, M, 2 * HFC, HFC, ACT_NONE, GX(wip ? wip + (size_t)HFC * HFC : nullptr)));
    WM_TRY(launch_mha16(h, s, PS, h->q16, HFC, h->kv16, 2 * HFC, h->kv16 + HFC, 2 * HFC, h->aoh16, HFC, B, HFC_HEADS,
                        HFC / HFC_HEADS, T, T, 1));                                                                      // :500-503
    WM_TRY(launch_gemm16(h, s, PS, h->aoh16, W16(h, a + "cross_attn.out_proj.weight"), W32(h, a + "cross_attn.out_proj.bias"),
                         h->pt32, 0, h->y1, nullptr, M, HFC, HFC, ACT_NONE, GX(W16P(h, a + "cross_attn.out_proj.weight"))));                                       // + residual :504
    WM_TRY(launch_layernorm(h, s, PS, h->y1, W32(h, a + "norm1.weight"), W32(h, a + "norm1.bias"), 1e-5f, h->y1n32, h->y1n16, M, HFC));
    WM_TRY(launch_gemm16(h, s, PS, h->y1n16, W16(h, a + "linear1.weight"), W32(h, a + "linear1.bias"), nullptr, 0, nullptr,
                         h->h1_16, M, HFC, HFC, ACT_RELU, GX(W16P(h, a + "linear1.weight"))));
    WM_TRY(launch_gemm16(h, s, PS, h->h1_16, W16(h, a + "linear2.weight"), W32(h, a + "linear2.bias"), h->y1n32, 0, h->z32,
                         nullptr, M, HFC, HFC, ACT_NONE, GX(W16P(h, a + "linear2.weight"))));                                                          // :506-508
    WM_TRY(launch_layernorm(h, s, PS, h->z32, W32(h, a + "norm2.weight"), W32(h, a + "norm2.bias"), 1e-5f, nullptr, h->y2_16, M, HFC));
    // scramble (:512): per tile [4096 tok,1024 ch] re-read as [1024, 4096]; make it the K-contiguous A operand
    WM_TRY(launch_simple(h, s, B * 16.8e6, transpose16_kernel, dim3(T / 64, HFC / 64, B), dim3(256), (const u16*)h->y2_16, (u16*)h->y2t16, HFC, T));
    // x = proj_back(scrambled) + t   (:513-514, :131)
    // Folded LayerNorm (WM_CFG_FOLD_LN; gemm16_v5.h "Folded LayerNorm"): a block whose qkv and lin1 run the 256-row-tile
    // 16-bit kernel takes its two LayerNorms inside those GEMMs.  raw_prec: the 16-bit type in which xn16 holds the copy of the
    // CURRENT residual stream (LDS-image order) with fold_stats its per-row partial statistics, or -1.
    auto fold_block = [&](int i) {
        const int pb = block_prec(h, i);
        const std::string b = e + "blocks." + std::to_string(i) + ".";
        return h->fold && (pb == WM_PREC_FP16 || (pb == WM_PREC_BF16 && h->fold_bf16)) && gemm16_takes_v5(M, 3 * D, D) && gemm16_takes_v5(M, 4 * D, D) &&
               h->wfold.count(b + "attn.qkv.weight") && h->wfold.count(b + "mlp.lin1.weight");
    };
    // Where the residual stream lives.  st_split: as the two 16-bit planes (xn16 = hi of type raw_prec, lo16), fp32 `resid` stale;
    // otherwise in `resid` (fp32), with xn16 / fold_stats its hi plane and statistics iff raw_prec >= 0.  The split form is used by
    // a call whose residual GEMMs run the 256-row-tile kernel (4+ tiles for ViT-H); a smaller call keeps fp32 and rounds the stream
    // to hi + lo in ln_stats_x16_kernel, so both forms carry the same values bit for bit (gemm16_v5.h "Split stream").
    const bool split_call = h->split && gemm16_takes_v5(M, D, D) && gemm16_takes_v5(M, D, 4 * D);
    // st_rows (fp8 blocks): as two planes of rows (x16last = hi bf16, lo16; columns at plane_pos), `resid` stale (gemm8.h PLANES).
    bool st_split = false, st_rows = false;
    int raw_prec = -1;
    auto to_fp32 = [&]() -> int {                           // planes -> resid (type boundaries, non-folded blocks, the bf16 neck input)
        if (st_split) WM_TRY(launch_stream_merge(h, s, raw_prec, h->xn16, h->lo16, h->resid, M, D));
        if (st_rows) WM_TRY(launch_stream_rows(h, s, WM_PREC_BF16, h->resid, h->x16last, h->lo16, M, D, true));
        st_split = st_rows = false;
        return 0;
    };
    auto planes_from_fp32 = [&](int P) -> int {             // resid -> statistics + planes of type P (resid rounded in place unless the call is split)
        WM_TRY(to_fp32());
        WM_TRY(launch_ln_stats16(h, s, P, h->resid, h->fold_stats, h->xn16, M, D, h->lo16, split_call ? nullptr : h->resid, h->overflow));
        raw_prec = P;
        st_split = split_call;
        return 0;
    };
    auto tap = [&](int which) -> int {
        if (h->tap_which != which) return 0;
        if (!st_split && !st_rows) return do_tap(h, s, which, B);
        WM_TRY(tap_alloc(h));
        if (st_rows) return launch_stream_rows(h, s, WM_PREC_BF16, h->tap_buf, h->x16last, h->lo16, M, D, true);
        return launch_stream_merge(h, s, raw_prec, h->xn16, h->lo16, h->tap_buf, M, D);
    };
    // a residual GEMM of a folded block of type P: x += A W^T + b, leaving the stream with planes + statistics of type P
    auto residual_gemm = [&](int P, const void* A, const std::string& wn, int K, int a_packed) -> int {
        GemmExtra x = GX(W16P(h, wn + ".weight"), a_packed);
        if (st_split) {                                     // planes in, planes out (in place), statistics out
            x.st_stats = h->fold_stats; x.res_hi = h->xn16; x.res_lo = h->lo16; x.out_lo = h->lo16; x.overflow = h->overflow;
            return launch_gemm16(h, s, P, A, W16(h, wn + ".weight"), W32(h, wn + ".bias"), nullptr, 0, nullptr, h->xn16, M, D, K, ACT_NONE, x);
        }
        const bool v5 = gemm16_takes_v5(M, D, K);
        if (v5 && !h->split) {                              // fp32 stream (WM_STREAM_SPLIT=0): statistics + 16-bit copy from the GEMM, as in round 3
            x.st_stats = h->fold_stats;
            WM_TRY(launch_gemm16(h, s, P, A, W16(h, wn + ".weight"), W32(h, wn + ".bias"), h->resid, 0, h->resid, h->xn16, M, D, K, ACT_NONE, x));
            raw_prec = P;
            return 0;
        }
        // half-width launch (1-2 tiles per call), or a call whose proj and lin2 disagree about the kernel: fp32 in place, then the
        // standalone statistics kernel, which also rounds the stream to hi + lo
        WM_TRY(launch_gemm16(h, s, P, A, W16(h, wn + ".weight"), W32(h, wn + ".bias"), h->resid, 0, h->resid, nullptr, M, D, K, ACT_NONE, x));
        raw_prec = -1;
        return planes_from_fp32(P);
    };
    {
        GemmExtra xb = GX(W16P(h, a + "proj_back.weight"));
        const bool produce = fold_block(0) && block_prec(h, 0) == PS && gemm16_takes_v5(M, D, HFC) && (split_call || !h->split);
        if (produce) { xb.st_stats = h->fold_stats; xb.overflow = h->overflow; if (split_call) xb.out_lo = h->lo16; }
        WM_TRY(launch_gemm16(h, s, PS, h->y2t16, W16(h, a + "proj_back.weight"), W32(h, a + "proj_back.bias"), h->tokbase, 0,
                             (produce && split_call) ? nullptr : h->resid, produce ? h->xn16 : nullptr, M, D, HFC, ACT_NONE, xb));
        if (produce) { raw_prec = PS; st_split = split_call; }
    }
    WM_TRY(tap(-1));

    // ---- transformer blocks (image_encoder.py:188-204) ----
    // x = x + proj(attn(norm1 x)); x = x + lin2(gelu(lin1(norm2 x))).
    // Per GEMM the operand type is the block's (block_prec) or, in fp8 mode, e4m3 for the GEMMs the handle's fp8 mask names
    // (WM_FP8_QKV | WM_FP8_PROJ | WM_FP8_MLP; lin1 and lin2 go together because lin1's epilogue writes lin2's operand) and
    // bf16 for the rest and for attention.  Each producer writes its consumer's operand type directly (LayerNorm / attention /
    // GELU epilogue -> e4m3 bytes or 16-bit), so no conversion pass exists in any mix.
    bool xn_packed = false;                                 // xn16 (a LayerNorm's output) is in LDS-image order
    for (int i = 0; i < h->depth; ++i) {
        const std::string b = e + "blocks." + std::to_string(i) + ".";
        const int PB = block_prec(h, i);
        const bool f8 = PB == WM_PREC_FP8;
        const int P = f8 ? WM_PREC_BF16 : PB;               // 16-bit type of this block (attention, non-fp8 GEMMs)
        const bool q8 = f8 && (h->fp8_gemms & WM_FP8_QKV), p8 = f8 && (h->fp8_gemms & WM_FP8_PROJ), m8 = f8 && (h->fp8_gemms & WM_FP8_MLP);
        auto W8 = [&](const std::string& n) { return h->w8.at(n); };
        // Activations that feed a 16-bit GEMM on the 256-row-tile kernel are written in LDS-image order by their producer
        // (gemm16_v5.h "Operand layout"): norm1 -> qkv, norm2 -> lin1, lin1's GELU epilogue -> lin2.  (proj's operand, the
        // attention output, stays row-major: a head's 80 columns do not fall on the 32-column pieces.)
        const bool pk_qkv = !h->row_major && !q8 && gemm16_takes_v5(M, 3 * D, D), pk_lin1 = !h->row_major && !m8 && gemm16_takes_v5(M, 4 * D, D);
        const bool pk_lin2 = pk_lin1 && gemm16_takes_v5(M, D, 4 * D);
        if (fold_block(i)) {
            // ---- both LayerNorms folded: statistics (and the stream's hi plane = the operand) from the producing residual GEMM, or
            // from the standalone kernel where that one is a half-width launch or of another operand type; normalisation in the
            // consuming GEMM's epilogue ----
            auto folded = [&](const std::string& wn, int act, int out_packed, void* out, int N) {
                GemmExtra x = GX(h->wfold.at(wn), 1, out_packed);
                x.fold_stats = h->fold_stats; x.fold_c1 = h->fold_c1.at(wn); x.fold_eps = 1e-6f;
                return launch_gemm16(h, s, P, h->xn16, W16(h, wn), h->fold_c2.at(wn), nullptr, 0, nullptr, out, M, N, D, act, x);
            };
            if (raw_prec != P) WM_TRY(planes_from_fp32(P));
            WM_TRY(sat_check(h, s, WM_SAT_LN, h->xn16, (int64_t)M * D, P));
            WM_TRY(folded(b + "attn.qkv.weight", ACT_NONE, 0, h->qkv16, 3 * D));
            WM_TRY(sat_check(h, s, WM_SAT_QKV, h->qkv16, (int64_t)M * 3 * D, P));
            WM_TRY(launch_encoder_attention(h, s, P, h->qkv16, W32(h, b + "attn.qkv.bias"), W32(h, b + "attn.rel_pos_h"),
                                            W32(h, b + "attn.rel_pos_w"), h->ao16, B, h->heads, h->hd, h->is_global[i] ? 0 : 14, nullptr, nullptr, nullptr, 0, 1));
            WM_TRY(sat_check(h, s, WM_SAT_ATTN, h->ao16, (int64_t)M * D, P));
            WM_TRY(residual_gemm(P, h->ao16, b + "attn.proj", D, 0));
            WM_TRY(sat_check(h, s, WM_SAT_LN, h->xn16, (int64_t)M * D, P));
            WM_TRY(folded(b + "mlp.lin1.weight", ACT_GELU, pk_lin2, h->hid16, 4 * D));
            WM_TRY(sat_check(h, s, WM_SAT_HID, h->hid16, (int64_t)M * 4 * D, P));
            WM_TRY(residual_gemm(P, h->hid16, b + "mlp.lin2", 4 * D, pk_lin2));
            WM_TRY(tap(i));
            continue;
        }
        // a block whose four GEMMs take e4m3 keeps the stream as planes of rows: proj / lin2 move the same 8 bytes per element, the
        // two LayerNorm passes read the 2-byte hi plane instead of 4-byte rows (their e4m3 output has a 2^-4 step; hi is bf16, 2^-9)
        const bool rows_blk = h->rows8 && q8 && p8 && m8 && D % 256 == 0 && D >= 512 && D <= 1536 && h->w8k.count(b + "attn.qkv.weight") && h->w8k.count(b + "mlp.lin1.weight");
        if (rows_blk && !st_rows) {
            WM_TRY(to_fp32());
            WM_TRY(launch_stream_rows(h, s, P, h->resid, h->x16last, h->lo16, M, D, false));
            st_rows = true;
        } else if (!rows_blk) {
            WM_TRY(to_fp32());
        }
        raw_prec = -1;
        if (st_rows) WM_TRY(launch_layernorm_plane8(h, s, P, h->x16last, W32(h, b + "norm1.weight"), W32(h, b + "norm1.bias"), 1e-6f, h->xn16, M, D));
        else WM_TRY(launch_layernorm_block(h, s, q8 ? WM_PREC_FP8 : P, h->resid, W32(h, b + "norm1.weight"), W32(h, b + "norm1.bias"), 1e-6f, h->xn16, M, D, pk_qkv));
        xn_packed = pk_qkv;
        WM_TRY(sat_check(h, s, WM_SAT_LN, h->xn16, (int64_t)M * D, q8 ? WM_PREC_FP8 : P));
        if (q8)
            WM_TRY(launch_gemm8(h, s, P, h->xn16, st_rows ? h->w8k.at(b + "attn.qkv.weight") : W8(b + "attn.qkv.weight"), W32(h, b + "attn.qkv.weight.wscale"), W32(h, b + "attn.qkv.bias"),
                                nullptr, nullptr, h->qkv16, nullptr, M, 3 * D, D, ACT_NONE));
        else
            WM_TRY(launch_gemm16(h, s, P, h->xn16, W16(h, b + "attn.qkv.weight"), W32(h, b + "attn.qkv.bias"), nullptr, 0, nullptr,
                                 h->qkv16, M, 3 * D, D, ACT_NONE, GX(W16P(h, b + "attn.qkv.weight"), xn_packed)));
        WM_TRY(sat_check(h, s, WM_SAT_QKV, h->qkv16, (int64_t)M * 3 * D, P));
        // the attention kernels write their output as e4m3 when proj consumes e4m3
        WM_TRY(launch_encoder_attention(h, s, P, h->qkv16, W32(h, b + "attn.qkv.bias"), W32(h, b + "attn.rel_pos_h"),
                                        W32(h, b + "attn.rel_pos_w"), h->ao16, B, h->heads, h->hd, h->is_global[i] ? 0 : 14, p8 ? h->ao8 : nullptr, nullptr, nullptr, 0, 1));
        if (p8) WM_TRY(sat_check(h, s, WM_SAT_ATTN, h->ao8, (int64_t)M * D, WM_PREC_FP8));
        else WM_TRY(sat_check(h, s, WM_SAT_ATTN, h->ao16, (int64_t)M * D, P));
        if (st_rows) {
            WM_TRY(launch_gemm8(h, s, P, h->ao8, W8(b + "attn.proj.weight"), W32(h, b + "attn.proj.weight.wscale"), W32(h, b + "attn.proj.bias"),
                                nullptr, nullptr, nullptr, nullptr, M, D, D, ACT_NONE, h->x16last, h->lo16));
        } else if (p8) {
            WM_TRY(launch_gemm8(h, s, P, h->ao8, W8(b + "attn.proj.weight"), W32(h, b + "attn.proj.weight.wscale"), W32(h, b + "attn.proj.bias"),
                                h->resid, h->resid, nullptr, nullptr, M, D, D, ACT_NONE));
        } else {
            WM_TRY(launch_gemm16(h, s, P, h->ao16, W16(h, b + "attn.proj.weight"), W32(h, b + "attn.proj.bias"), h->resid, 0,
                                 h->resid, nullptr, M, D, D, ACT_NONE, GX(W16P(h, b + "attn.proj.weight"))));
        }
        if (st_rows) WM_TRY(launch_layernorm_plane8(h, s, P, h->x16last, W32(h, b + "norm2.weight"), W32(h, b + "norm2.bias"), 1e-6f, h->xn16, M, D));
        else WM_TRY(launch_layernorm_block(h, s, m8 ? WM_PREC_FP8 : P, h->resid, W32(h, b + "norm2.weight"), W32(h, b + "norm2.bias"), 1e-6f, h->xn16, M, D, pk_lin1));
        xn_packed = pk_lin1;
        WM_TRY(sat_check(h, s, WM_SAT_LN, h->xn16, (int64_t)M * D, m8 ? WM_PREC_FP8 : P));
        if (m8) {
            WM_TRY(launch_gemm8(h, s, P, h->xn16, st_rows ? h->w8k.at(b + "mlp.lin1.weight") : W8(b + "mlp.lin1.weight"), W32(h, b + "mlp.lin1.weight.wscale"), W32(h, b + "mlp.lin1.bias"),
                                nullptr, nullptr, nullptr, h->hid16, M, 4 * D, D, ACT_GELU));
            WM_TRY(sat_check(h, s, WM_SAT_HID, h->hid16, (int64_t)M * 4 * D, WM_PREC_FP8));
            if (st_rows)
                WM_TRY(launch_gemm8(h, s, P, h->hid16, W8(b + "mlp.lin2.weight"), W32(h, b + "mlp.lin2.weight.wscale"), W32(h, b + "mlp.lin2.bias"),
                                    nullptr, nullptr, nullptr, nullptr, M, D, 4 * D, ACT_NONE, h->x16last, h->lo16));
            else
                WM_TRY(launch_gemm8(h, s, P, h->hid16, W8(b + "mlp.lin2.weight"), W32(h, b + "mlp.lin2.weight.wscale"), W32(h, b + "mlp.lin2.bias"),
                                    h->resid, h->resid, nullptr, nullptr, M, D, 4 * D, ACT_NONE));
        } else {
            WM_TRY(launch_gemm16(h, s, P, h->xn16, W16(h, b + "mlp.lin1.weight"), W32(h, b + "mlp.lin1.bias"), nullptr, 0, nullptr,
                                 h->hid16, M, 4 * D, D, ACT_GELU, GX(W16P(h, b + "mlp.lin1.weight"), xn_packed, pk_lin2)));
            WM_TRY(sat_check(h, s, WM_SAT_HID, h->hid16, (int64_t)M * 4 * D, P));
            WM_TRY(launch_gemm16(h, s, P, h->hid16, W16(h, b + "mlp.lin2.weight"), W32(h, b + "mlp.lin2.bias"), h->resid, 0,
                                 h->resid, nullptr, M, D, 4 * D, ACT_NONE, GX(W16P(h, b + "mlp.lin2.weight"), pk_lin2)));
        }
        WM_TRY(tap(i));
    }

    // ---- neck (image_encoder.py:105-121,136) ----
    // the neck's operand is fp16(x).  With the split stream and fp16 blocks that is the hi plane itself (LDS-image order: the
    // 256-row-tile kernel takes it as it is, a half-width launch gets it unpacked); otherwise fp16 of the fp32 stream.
    const void* neck_a = h->x16last;
    int neck_packed = 0;
    if (h->split && raw_prec == PS) {
        if (gemm16_takes_v5(M, OUTC, D)) { neck_a = h->xn16; neck_packed = 1; }
        else WM_TRY(launch_simple(h, s, B * 21.0e6, unpack16_lds_image_kernel, dim3(grid_for((int64_t)M * D / 8)), dim3(256), (const uint4*)h->xn16, (uint4*)h->x16last, (int64_t)M, D));
    } else if (st_rows) {                                   // the fp8 blocks' planes: one pass to fp16 rows (hid16 is free after the last block)
        neck_a = h->hid16;
        WM_TRY(launch_simple(h, s, B * 31.5e6, stream_rows_to_fp16_kernel<BF16>, dim3(grid_for((int64_t)M * D / 4)), dim3(256), (const u16*)h->x16last, (const u16*)h->lo16,
                             (u16*)h->hid16, (int64_t)M * D / 4, D));
        st_rows = false;
    } else {
        WM_TRY(to_fp32());
        WM_TRY(launch_simple(h, s, B * 31.5e6, cvt_f32_to_16_kernel<FP16>, dim3(grid_for((int64_t)M * D / 4)), dim3(256), (const float*)h->resid, (u16*)h->x16last, (int64_t)M * D / 4));
    }
    WM_TRY(sat_check(h, s, WM_SAT_LAST, neck_a, (int64_t)M * D, PS));
    WM_TRY(launch_gemm16(h, s, PS, neck_a, W16(h, e + "neck.0.weight"), nullptr, nullptr, 0, h->n1, nullptr, M, OUTC, D, ACT_NONE, GX(W16P(h, e + "neck.0.weight"), neck_packed)));
    WM_TRY(launch_layernorm(h, s, PS, h->n1, W32(h, e + "neck.1.weight"), W32(h, e + "neck.1.bias"), 1e-6f, nullptr, h->n1n16, M, OUTC));
    WM_TRY(launch_conv3x3_16(h, s, PS, h->n1n16, W16(h, e + "neck.2.weight"), h->n2, M, OUTC, OUTC));
    WM_TRY(launch_layernorm(h, s, PS, h->n2, W32(h, e + "neck.3.weight"), W32(h, e + "neck.3.bias"), 1e-6f, h->emb_nhwc, nullptr, M, OUTC));
    if (out_nchw)
        WM_TRY(launch_simple(h, s, B * 8.4e6, transpose32_kernel, dim3(OUTC / 64, T / 64, B), dim3(256), (const float*)h->emb_nhwc, out_nchw, T, OUTC));
    return 0;
}

// attention block of the decoder (transformer.py:217-240) on fp32 buffers
struct DecAttnW { const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo; int internal; };

DecAttnW dec_w(wm_handle* h, const std::string& p, int internal) {
    return DecAttnW{W32(h, p + "q_proj.weight"), W32(h, p + "q_proj.bias"), W32(h, p + "k_proj.weight"), W32(h, p + "k_proj.bias"),
                    W32(h, p + "v_proj.weight"), W32(h, p + "v_proj.bias"), W32(h, p + "out_proj.weight"), W32(h, p + "out_proj.bias"), internal};
}

}  // namespace

// elementwise fp32 add with a row-broadcast second operand: out[r,c] = a[r,c] + b[r % mod, c]
__global__ __launch_bounds__(256) void add_bcast_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                        int64_t rows, int C, int mod) {
    const int64_t n4 = rows * C / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = (i * 4) / C;
        const int c = (int)((i * 4) % C);
        const f32x4 y = *(const f32x4*)(b + (r % mod) * C + c);
        *(f32x4*)(out + i * 4) = a ? *(const f32x4*)(a + i * 4) + y : y;        // a == null: the rows of b repeated
    }
}

namespace {

int decoder_impl(wm_handle* h, const float* keys_nhwc, float* logits, float* boxes, int B, hipStream_t s) {
    const std::string t = "mask_decoder.transformer.";
    const int E = OUTC, Mk = B * T, Mq = B * NQ;
    const float* tok = W32(h, "mask_decoder.mask_tokens.weight");      // [51,256] = tokens and query PE (box_decoder.py:128-131)
    float* keys = h->dkeys;
    HIP_TRY(hipMemcpyAsync(keys, keys_nhwc, (size_t)Mk * E * 4, hipMemcpyDeviceToDevice, s));
    float* queries = h->dq;
    // every tile starts from the same 51 tokens (one launch; it was one copy per tile)
    WM_TRY(launch_simple(h, s, 0.0, add_bcast_kernel, dim3(grid_for((int64_t)Mq * E / 4)), dim3(256), (const float*)nullptr, tok, queries, (int64_t)Mq, E, NQ));

    auto add_q = [&](float* out) {   // queries + query_pe
        return launch_simple(h, s, 0.0, add_bcast_kernel, dim3(grid_for((int64_t)Mq * E / 4)), dim3(256), (const float*)queries, tok, out, (int64_t)Mq, E, NQ);
    };
    auto add_k = [&](float* out) {   // keys + key_pe
        return launch_simple(h, s, 0.0, add_bcast_kernel, dim3(grid_for((int64_t)Mk * E / 4)), dim3(256), (const float*)keys, (const float*)h->kpe, out, (int64_t)Mk, E, T);
    };
    auto ln = [&](float* x, const std::string& n, int rows) {
        return launch_layernorm(h, s, WM_PREC_FP16, x, W32(h, n + ".weight"), W32(h, n + ".bias"), 1e-5f, x, nullptr, rows, E);   // fp32 in place: the 16-bit type is unused
    };
    // token -> image attention: q from (queries+pe), k from (keys+pe) [kin], v from keys; result added to queries
    auto token_to_image = [&](const DecAttnW& w, const float* kin) -> int {
        WM_TRY(add_q(h->dt_h1));
        WM_TRY(launch_gemm32(h, s, h->dt_h1, w.wq, w.bq, nullptr, h->dt_q, Mq, w.internal, E, ACT_NONE));
        WM_TRY(launch_gemm32(h, s, kin, w.wk, w.bk, nullptr, h->dk_a, Mk, w.internal, E, ACT_NONE));
        WM_TRY(launch_gemm32(h, s, keys, w.wv, w.bv, nullptr, h->dk_b, Mk, w.internal, E, ACT_NONE));
        WM_TRY(launch_mha32(h, s, h->dt_q, h->dk_a, h->dk_b, h->dt_att, B, 8, w.internal / 8, NQ, T));
        WM_TRY(launch_gemm32(h, s, h->dt_att, w.wo, w.bo, queries, queries, Mq, E, w.internal, ACT_NONE));
        return 0;
    };
    float* kpe_sum = h->n1;   // reuse [B*T,256] fp32 scratch of the neck: keys + key_pe

    for (int i = 0; i < 2; ++i) {
        const std::string L = t + "layers." + std::to_string(i) + ".";
        // (1) self attention of the tokens (transformer.py:151-158)
        {
            const DecAttnW w = dec_w(h, L + "self_attn.", E);
            const float* qin = queries;
            if (i != 0) { WM_TRY(add_q(h->dt_h1)); qin = h->dt_h1; }
            WM_TRY(launch_gemm32(h, s, qin, w.wq, w.bq, nullptr, h->dt_q, Mq, E, E, ACT_NONE));
            WM_TRY(launch_gemm32(h, s, qin, w.wk, w.bk, nullptr, h->dt_k, Mq, E, E, ACT_NONE));
            WM_TRY(launch_gemm32(h, s, queries, w.wv, w.bv, nullptr, h->dt_v, Mq, E, E, ACT_NONE));
            WM_TRY(launch_mha32(h, s, h->dt_q, h->dt_k, h->dt_v, h->dt_att, B, 8, E / 8, NQ, NQ));
            // layer 0 replaces the queries (no residual, :155-156); later layers add
            WM_TRY(launch_gemm32(h, s, h->dt_att, w.wo, w.bo, i == 0 ? nullptr : queries, queries, Mq, E, E, ACT_NONE));
            WM_TRY(ln(queries, L + "norm1", Mq));
        }
        // (2) tokens attend to the image (:160-165)
        WM_TRY(add_k(kpe_sum));
        WM_TRY(token_to_image(dec_w(h, L + "cross_attn_token_to_image.", E / 2), kpe_sum));
        WM_TRY(ln(queries, L + "norm2", Mq));
        // (3) MLP (:167-170)
        WM_TRY(launch_gemm32(h, s, queries, W32(h, L + "mlp.lin1.weight"), W32(h, L + "mlp.lin1.bias"), nullptr, h->dt_hid, Mq, DEC_MLP, E, ACT_RELU));
        WM_TRY(launch_gemm32(h, s, h->dt_hid, W32(h, L + "mlp.lin2.weight"), W32(h, L + "mlp.lin2.bias"), queries, queries, Mq, E, DEC_MLP, ACT_NONE));
        WM_TRY(ln(queries, L + "norm3", Mq));
        // (4) image attends to the tokens (:172-178): q = keys+pe, k = queries+pe, v = queries
        {
            const DecAttnW w = dec_w(h, L + "cross_attn_image_to_token.", E / 2);
            WM_TRY(add_q(h->dt_h1));
            WM_TRY(launch_gemm32(h, s, kpe_sum, w.wq, w.bq, nullptr, h->dk_a, Mk, w.internal, E, ACT_NONE));
            WM_TRY(launch_gemm32(h, s, h->dt_h1, w.wk, w.bk, nullptr, h->dt_k, Mq, w.internal, E, ACT_NONE));
            WM_TRY(launch_gemm32(h, s, queries, w.wv, w.bv, nullptr, h->dt_v, Mq, w.internal, E, ACT_NONE));
            WM_TRY(launch_mha32(h, s, h->dk_a, h->dt_k, h->dt_v, h->dk_c, B, 8, w.internal / 8, T, NQ));
            WM_TRY(launch_gemm32(h, s, h->dk_c, w.wo, w.bo, keys, keys, Mk, E, w.internal, ACT_NONE));
            WM_TRY(ln(keys, L + "norm4", Mk));
        }
    }
    // final token -> image attention (transformer.py:100-104)
    WM_TRY(add_k(kpe_sum));
    WM_TRY(token_to_image(dec_w(h, t + "final_attn_token_to_image.", E / 2), kpe_sum));
    WM_TRY(ln(queries, t + "norm_final_attn", Mq));

    // heads (box_decoder.py:102-103)
    const std::string c = "mask_decoder.class_embed.layers.", bb = "mask_decoder.bbox_embed.layers.";
    WM_TRY(launch_gemm32(h, s, queries, W32(h, c + "0.weight"), W32(h, c + "0.bias"), nullptr, h->dt_h1, Mq, E, E, ACT_RELU));
    WM_TRY(launch_gemm32(h, s, h->dt_h1, W32(h, c + "1.weight"), W32(h, c + "1.bias"), nullptr, h->dt_h2, Mq, E, E, ACT_RELU));
    WM_TRY(launch_gemm32(h, s, h->dt_h2, W32(h, c + "2.weight"), W32(h, c + "2.bias"), nullptr, logits, Mq, WM_NUM_LOGITS, E, ACT_NONE));
    WM_TRY(launch_gemm32(h, s, queries, W32(h, bb + "0.weight"), W32(h, bb + "0.bias"), nullptr, h->dt_h1, Mq, E, E, ACT_RELU));
    WM_TRY(launch_gemm32(h, s, h->dt_h1, W32(h, bb + "1.weight"), W32(h, bb + "1.bias"), nullptr, h->dt_h2, Mq, E, E, ACT_RELU));
    WM_TRY(launch_gemm32(h, s, h->dt_h2, W32(h, bb + "2.weight"), W32(h, bb + "2.bias"), nullptr, boxes, Mq, 4, E, ACT_SIGMOID));
    return 0;
}

}  // namespace

extern "C" int wm_hfc_fft(wm_handle* h, const float* x_dev, float* hfc_dev, int batch, void* stream) {
    if (h && !h->finalized) { /* the FFT needs no weights */ }
    if (!h) return fail("wm_hfc_fft: null handle");
    if (batch <= 0 || batch > h->maxB) return fail("wm_hfc_fft: batch %d outside 1..%d", batch, h->maxB);
    if (!x_dev || !hfc_dev) return fail("wm_hfc_fft: null buffer");
    HIP_TRY(hipSetDevice(h->device));
    return fft_impl(h, x_dev, hfc_dev, batch, (hipStream_t)stream);
}

extern "C" int wm_encoder_forward(wm_handle* h, const float* x_dev, const float* hfc_dev, float* out_dev, int batch, void* stream) {
    WM_TRY(check_ready(h, batch, "wm_encoder_forward", true, false));
    if (!x_dev || !hfc_dev || !out_dev) return fail("wm_encoder_forward: null buffer");
    WM_TRY(encoder_impl(h, x_dev, hfc_dev, out_dev, batch, (hipStream_t)stream));
    // opt-in fused GEMM + LayerNorm: a partner time-out means wrong numbers, so it must never pass silently; the check
    // reads the flag on the launch stream (one stream synchronisation per call, paid only with the option on)
    return 0;
}

extern "C" int wm_decoder_forward(wm_handle* h, const float* emb_dev, float* logits_dev, float* boxes_dev, int batch, void* stream) {
    WM_TRY(check_ready(h, batch, "wm_decoder_forward", false, true));
    if (!emb_dev || !logits_dev || !boxes_dev) return fail("wm_decoder_forward: null buffer");
    hipStream_t s = (hipStream_t)stream;
    // NCHW (B,256,4096) -> token-major (B,4096,256) (transformer.py:83)
    WM_TRY(launch_simple(h, s, batch * 8.4e6, transpose32_kernel, dim3(T / 64, OUTC / 64, batch), dim3(256), emb_dev, h->emb_nhwc, OUTC, T));
    return decoder_impl(h, h->emb_nhwc, logits_dev, boxes_dev, batch, s);
}

extern "C" int wm_postprocess_nms(wm_handle* h, const float* logits_dev, const float* boxes_dev, const float* target_sizes_dev,
                                  float conf_thr, float score_thr, float iou_thr, wm_box_record* records_dev, int batch, void* stream) {
    if (batch <= 0) return fail("wm_postprocess_nms: batch %d", batch);
    if (!logits_dev || !boxes_dev || !target_sizes_dev || !records_dev) return fail("wm_postprocess_nms: null buffer");
    if (h) HIP_TRY(hipSetDevice(h->device));     // weightless kernel: a NULL handle launches on the current device
    return launch_simple(h, (hipStream_t)stream, 0.0, postprocess_nms_kernel, dim3(batch), dim3(64), logits_dev, boxes_dev,
                         target_sizes_dev, conf_thr, score_thr, iou_thr, records_dev);
}

extern "C" int wm_forward(wm_handle* h, const float* x_dev, const float* target_sizes_dev, float* logits_dev, float* boxes_dev,
                          wm_box_record* records_dev, int batch, void* stream) {
    WM_TRY(check_ready(h, batch, "wm_forward", true, true));
    if (!x_dev) return fail("wm_forward: null input");
    hipStream_t s = (hipStream_t)stream;
    WM_TRY(fft_impl(h, x_dev, h->hfc, batch, s, true));                                    // network.py:61 (+ the embeds' 16-bit operands)
    WM_TRY(encoder_impl(h, x_dev, h->hfc, nullptr, batch, s, true));                       // network.py:65
    WM_TRY(decoder_impl(h, h->emb_nhwc, h->logits, h->boxes, batch, s));                   // network.py:79-86
    const float* ts = target_sizes_dev ? target_sizes_dev : h->tsz_default;
    WM_TRY(launch_simple(h, s, 0.0, postprocess_nms_kernel, dim3(batch), dim3(64), (const float*)h->logits, (const float*)h->boxes, ts,
                         0.05f, 0.5f, 0.4f, h->records));
    if (logits_dev) HIP_TRY(hipMemcpyAsync(logits_dev, h->logits, (size_t)batch * NQ * WM_NUM_LOGITS * 4, hipMemcpyDeviceToDevice, s));
    if (boxes_dev) HIP_TRY(hipMemcpyAsync(boxes_dev, h->boxes, (size_t)batch * NQ * 16, hipMemcpyDeviceToDevice, s));
    if (records_dev) HIP_TRY(hipMemcpyAsync(records_dev, h->records, (size_t)batch * NQ * sizeof(wm_box_record), hipMemcpyDeviceToDevice, s));
    return 0;
}

// ---------------------------------------------------------------------------
// taps / profiling
// ---------------------------------------------------------------------------
extern "C" int wm_set_tap(wm_handle* h, int which) {
    if (!h) return fail("wm_set_tap: null handle");
    if (which < -3 || which >= h->depth) return fail("wm_set_tap: %d out of range", which);
    h->tap_which = which;
    return 0;
}

extern "C" int wm_read_tap(wm_handle* h, float* out_dev, int batch, void* stream) {
    if (!h || !out_dev) return fail("wm_read_tap: null argument");
    if (!h->tap_buf) return fail("wm_read_tap: no tap captured");
    if (batch <= 0 || batch > h->maxB) return fail("wm_read_tap: bad batch");
    HIP_TRY(hipMemcpyAsync(out_dev, h->tap_buf, (size_t)batch * T * h->D * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int wm_profile_enable(wm_handle* h, int on) {
    if (!h) return fail("wm_profile_enable: null handle");
    h->prof.on = on != 0;
    return 0;
}

extern "C" int wm_profile_reset(wm_handle* h) {
    if (!h) return fail("wm_profile_reset: null handle");
    WM_TRY(prof_collect(h));
    for (auto& a : h->prof.acc) a = wm_kclass_stat{};
    return 0;
}

extern "C" int wm_profile_read(wm_handle* h, wm_kclass_stat* out) {
    if (!h || !out) return fail("wm_profile_read: null argument");
    WM_TRY(prof_collect(h));
    for (int i = 0; i < WM_KCLASS_COUNT; ++i) out[i] = h->prof.acc[i];
    return 0;
}

extern "C" int wm_stream_overflow(wm_handle* h, int reset) {
    if (!h || !h->overflow) return fail("wm_stream_overflow: null handle");
    const int v0 = ((volatile int*)h->overflow)[0], v1 = ((volatile int*)h->overflow)[1];
    if (reset) ((volatile int*)h->overflow)[0] = ((volatile int*)h->overflow)[1] = 0;
    return (v0 != 0 ? WM_OVERFLOW_STREAM : 0) | (v1 != 0 ? WM_OVERFLOW_DECODER : 0);
}

extern "C" int wm_debug_saturation_enable(wm_handle* h, int on) {
    if (!h) return fail("wm_debug_saturation_enable: null handle");
    h->sat_on = on != 0;
    return 0;
}

extern "C" int wm_debug_saturation_read(wm_handle* h, int64_t* out, int n, int reset, void* stream) {
    if (!h || !out || n < WM_SAT_COUNT) return fail("wm_debug_saturation_read: need room for %d counters", WM_SAT_COUNT);
    HIP_TRY(hipSetDevice(h->device));
    unsigned long long host[WM_SAT_COUNT];
    HIP_TRY(hipMemcpyAsync(host, h->sat_counts, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < WM_SAT_COUNT; ++i) out[i] = (int64_t)host[i];
    if (reset) HIP_TRY(hipMemsetAsync(h->sat_counts, 0, sizeof(host), (hipStream_t)stream));
    return 0;
}

// ---------------------------------------------------------------------------
// single-op entry points
// ---------------------------------------------------------------------------
extern "C" int wm_op_cvt_f32_to_16(const float* in_dev, void* out_dev, int64_t n, int precision, void* stream) {
    if (n % 4) return fail("cvt: n must be a multiple of 4");
    if (precision == WM_PREC_FP16) hipLaunchKernelGGL(cvt_f32_to_16_kernel<FP16>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, in_dev, (u16*)out_dev, n / 4);
    else hipLaunchKernelGGL(cvt_f32_to_16_kernel<BF16>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, in_dev, (u16*)out_dev, n / 4);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int wm_op_cvt_16_to_f32(const void* in_dev, float* out_dev, int64_t n, int precision, void* stream) {
    if (precision == WM_PREC_FP16) hipLaunchKernelGGL(cvt_16_to_f32_kernel<FP16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const u16*)in_dev, out_dev, n);
    else hipLaunchKernelGGL(cvt_16_to_f32_kernel<BF16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const u16*)in_dev, out_dev, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int wm_preprocess_u8(const uint8_t* img_dev, float* out_dev, int batch, int height, int width, void* stream) {
    if (!img_dev || !out_dev) return fail("wm_preprocess_u8: null buffer");
    if (batch <= 0 || height <= 0 || width <= 0 || height > 1024 || width > 1024)
        return fail("wm_preprocess_u8: batch %d, %dx%d outside 1..1024 (larger images are cropped by the caller, utils/misc.py:57-60)", batch, height, width);
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for((int64_t)batch * 1024 * 256)), dim3(256), 0, (hipStream_t)stream,
                       img_dev, out_dev, batch, height, width);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- N3: large-frame tiling front end ----
extern "C" int wm_tile_frame_u8(const uint8_t* frame_dev, const int32_t* origins_dev, float* out_dev, int n_tiles, int height, int width,
                                void* stream) {
    if (!frame_dev || !origins_dev || !out_dev) return fail("wm_tile_frame_u8: null buffer");
    if (n_tiles <= 0 || height <= 0 || width <= 0) return fail("wm_tile_frame_u8: n_tiles %d, frame %dx%d", n_tiles, height, width);
    hipLaunchKernelGGL(tile_frame_u8_kernel, dim3(grid_for((int64_t)n_tiles * 1024 * 256)), dim3(256), 0, (hipStream_t)stream, frame_dev,
                       (const int*)origins_dev, out_dev, n_tiles, height, width);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int wm_merge_tiles_nms(const wm_box_record* records_dev, const int32_t* origins_dev, int n_tiles, float iou_thr,
                                  wm_box_record* merged_dev, void* stream) {
    if (!records_dev || !origins_dev || !merged_dev) return fail("wm_merge_tiles_nms: null buffer");
    const int n_slots = n_tiles * WM_NUM_QUERIES;
    if (n_tiles <= 0 || n_slots > MERGE_MAX_SLOTS) return fail("wm_merge_tiles_nms: %d tiles (1..%d)", n_tiles, MERGE_MAX_SLOTS / WM_NUM_QUERIES);
    constexpr int LDS = MERGE_MAX_SLOTS * (5 * 4 + 4 + 2) + 16;
    WM_TRY(set_max_lds((const void*)merge_tiles_nms_kernel, LDS));
    hipLaunchKernelGGL(merge_tiles_nms_kernel, dim3(1), dim3(MERGE_THREADS), LDS, (hipStream_t)stream, records_dev, (const int*)origins_dev, n_slots,
                       iou_thr, merged_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- N1: val transform resize (PIL bilinear semantics) + normalise + pad ----
namespace {

// augmentation.py:80-99 (get_size_with_aspect_ratio): (w, h), size, max_size -> (oh, ow)
void resized_size(int w, int h, int size, int max_size, int* oh, int* ow) {
    if (max_size > 0) {
        const double mn = (double)std::min(w, h), mx = (double)std::max(w, h);       // Python floats are doubles
        if (mx / mn * size > max_size) size = (int)std::nearbyint(max_size * mn / mx);               // Python round(): half to even
    }
    if ((w <= h && w == size) || (h <= w && h == size)) { *oh = h; *ow = w; return; }
    if (w < h) { *ow = size; *oh = (int)((double)size * h / w); }
    else { *oh = size; *ow = (int)((double)size * w / h); }
}

// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc, bilinear filter (support 1), whole axis; double arithmetic
// in the same operation order
void resize_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk, int& ksize) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> w((size_t)ksize + 2);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double v = (x + xmin - center + 0.5) * ss;
            if (v < 0.0) v = -v;
            const double wt = v < 1.0 ? 1.0 - v : 0.0;
            w[x] = wt;
            ww += wt;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) w[x] /= ww;
            kk[(size_t)xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << RESIZE_PREC_BITS)) : (int)(0.5 + w[x] * (1 << RESIZE_PREC_BITS));
        }
        bounds[(size_t)xx * 2] = xmin;
        bounds[(size_t)xx * 2 + 1] = xmax;
    }
}

struct ResizePlan {          // device tables of one geometry; owned by the library for the life of the process
    int oh = 0, ow = 0, ksx = 0, ksy = 0;
    int *bx = nullptr, *kx = nullptr, *by = nullptr, *ky = nullptr;
};
struct ResizeTmp { unsigned char* p = nullptr; size_t bytes = 0; };
struct ResizeDevState {
    std::map<std::array<int, 4>, ResizePlan> plans;      // (h, w, size, max_size); read-only once built
    // the intermediate (horizontally resampled) image, one per STREAM: calls on different streams of a device may overlap
    // on the GPU (a loader thread's side stream), and a shared buffer would be overwritten under the first call's kernels
    std::map<hipStream_t, ResizeTmp> tmp;
};
std::map<int, ResizeDevState> g_resize;

int upload_ints(const std::vector<int>& v, int** out) {
    HIP_TRY(hipMalloc((void**)out, v.size() * 4));
    HIP_TRY(hipMemcpy(*out, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

// host-only (tests): the coefficient tables the resize kernels use for one axis; ksize_out = taps per output, bounds [out][2], kk [out][ksize]
extern "C" int wm_debug_resize_coeffs(int in_size, int out_size, int* bounds_out, int* kk_out, int kk_capacity, int* ksize_out) {
    if (in_size <= 0 || out_size <= 0 || !bounds_out || !kk_out || !ksize_out) return fail("wm_debug_resize_coeffs: bad argument");
    std::vector<int> b, k;
    int ks = 0;
    resize_coeffs(in_size, out_size, b, k, ks);
    if ((size_t)kk_capacity < k.size()) return fail("wm_debug_resize_coeffs: need room for %zu coefficients", k.size());
    memcpy(bounds_out, b.data(), b.size() * 4);
    memcpy(kk_out, k.data(), k.size() * 4);
    *ksize_out = ks;
    return 0;
}

extern "C" int wm_resized_size(int height, int width, int size, int max_size, int* out_h, int* out_w) {
    if (height <= 0 || width <= 0 || size <= 0 || !out_h || !out_w) return fail("wm_resized_size: bad argument");
    resized_size(width, height, size, max_size, out_h, out_w);
    return 0;
}

extern "C" int wm_preprocess_u8_resized(const uint8_t* img_dev, float* out_dev, int batch, int height, int width, int size, int max_size,
                                        void* stream) {
    if (!img_dev || !out_dev) return fail("wm_preprocess_u8_resized: null buffer");
    if (batch <= 0 || height <= 0 || width <= 0 || size <= 0) return fail("wm_preprocess_u8_resized: batch %d, %dx%d, size %d", batch, height, width, size);
    int oh, ow;
    resized_size(width, height, size, max_size, &oh, &ow);
    if (oh > 1024 || ow > 1024 || oh <= 0 || ow <= 0)
        return fail("wm_preprocess_u8_resized: %dx%d resizes to %dx%d, outside the 1024x1024 canvas (utils/misc.py:57-60 crops; not built)", height, width, oh, ow);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_dev_mu);
    ResizeDevState& st = g_resize[dev];
    const std::array<int, 4> key{height, width, size, max_size};
    auto it = st.plans.find(key);
    if (it == st.plans.end()) {
        ResizePlan pl;
        pl.oh = oh; pl.ow = ow;
        std::vector<int> b, k;
        resize_coeffs(width, ow, b, k, pl.ksx);
        WM_TRY(upload_ints(b, &pl.bx)); WM_TRY(upload_ints(k, &pl.kx));
        resize_coeffs(height, oh, b, k, pl.ksy);
        WM_TRY(upload_ints(b, &pl.by)); WM_TRY(upload_ints(k, &pl.ky));
        it = st.plans.emplace(key, pl).first;
    }
    const ResizePlan& pl = it->second;
    const size_t need = (size_t)batch * height * ow * 3;
    hipStream_t s = (hipStream_t)stream;
    ResizeTmp& tmp = st.tmp[s];
    if (need > tmp.bytes) {
        if (tmp.p) HIP_TRY(hipFree(tmp.p));              // hipFree synchronises the device: no kernel still reads the old buffer
        tmp.p = nullptr; tmp.bytes = 0;
        HIP_TRY(hipMalloc((void**)&tmp.p, need));
        tmp.bytes = need;
    }
    // horizontal pass: the row-staged kernel where its geometry holds (<= 20 taps, <= 1024 output columns, a row fits LDS), else the generic one
    const int64_t rows_total = (int64_t)batch * height;
    const int lds_h = ((width * 3 + 3 + 3) / 4 + 1) * 4 + 64;      // row + alignment shift, + slack for the zero-coefficient taps (KMAX * 3 bytes)
    const bool fast_h = pl.ksx <= 20 && ow <= 1024 && lds_h <= 64 * 1024 && !(getenv("WM_RESIZE_GENERIC") && atoi(getenv("WM_RESIZE_GENERIC")));
    if (fast_h) {
        const int rpb = (int)std::max<int64_t>(4, std::min<int64_t>(16, rows_total / (256 * 8)));     // rows per workgroup: the coefficient registers are loaded once per workgroup
        const dim3 grid((unsigned)((rows_total + rpb - 1) / rpb));
        const int64_t in_bytes = rows_total * width * 3;
#define WM_RH(KM, OP) hipLaunchKernelGGL((resize_h_rows_kernel<KM, OP>), grid, dim3(256), lds_h, s, img_dev, tmp.p, (const int*)pl.bx, (const int*)pl.kx, \
                                         pl.ksx, rows_total, width, ow, rpb, in_bytes)
        const int opt = (ow + 255) / 256;
        if (pl.ksx <= 4) { if (opt <= 1) WM_RH(4, 1); else if (opt <= 2) WM_RH(4, 2); else if (opt <= 3) WM_RH(4, 3); else WM_RH(4, 4); }
        else if (pl.ksx <= 12) { if (opt <= 1) WM_RH(12, 1); else if (opt <= 2) WM_RH(12, 2); else if (opt <= 3) WM_RH(12, 3); else WM_RH(12, 4); }
        else { if (opt <= 1) WM_RH(20, 1); else if (opt <= 2) WM_RH(20, 2); else if (opt <= 3) WM_RH(20, 3); else WM_RH(20, 4); }
#undef WM_RH
    } else {
        hipLaunchKernelGGL(resize_h_u8_kernel, dim3(grid_for((int64_t)batch * height * ow)), dim3(256), 0, s, img_dev, tmp.p, (const int*)pl.bx,
                           (const int*)pl.kx, pl.ksx, batch, height, width, ow);
    }
    if (ow % 4 == 0 && !(getenv("WM_RESIZE_GENERIC") && atoi(getenv("WM_RESIZE_GENERIC"))))
        hipLaunchKernelGGL(resize_v_normalize4_kernel, dim3((unsigned)batch * 1024u), dim3(256), 0, s, (const unsigned char*)tmp.p, out_dev,
                           (const int*)pl.by, (const int*)pl.ky, pl.ksy, height, ow, oh);
    else
        hipLaunchKernelGGL(resize_v_normalize_kernel, dim3(grid_for((int64_t)batch * 1024 * 1024)), dim3(256), 0, s, (const unsigned char*)tmp.p, out_dev,
                           (const int*)pl.by, (const int*)pl.ky, pl.ksy, batch, height, ow, oh);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int wm_op_gemm16(const void* a_dev, const void* w_dev, const float* bias_dev, const float* residual_dev, int res_mod,
                            float* out_f32_dev, void* out_16_dev, int M, int N, int K, int act, int precision, void* stream) {
    const int layout = act & (WM_GEMM_W_PACKED | WM_GEMM_A_PACKED | WM_GEMM_OUT_PACKED);
    act &= ~(WM_GEMM_W_PACKED | WM_GEMM_A_PACKED | WM_GEMM_OUT_PACKED);
    if (layout && !gemm16_takes_v5(M, N, K))
        return fail("wm_op_gemm16: M=%d N=%d K=%d runs on a half-width kernel, which takes row-major operands only (wm_op_gemm16_takes_packed)", M, N, K);
    return launch_gemm16(nullptr, (hipStream_t)stream, precision, a_dev, (layout & WM_GEMM_W_PACKED) ? nullptr : w_dev, bias_dev, residual_dev, res_mod,
                         out_f32_dev, out_16_dev, M, N, K, act, GX((layout & WM_GEMM_W_PACKED) ? w_dev : nullptr, (layout & WM_GEMM_A_PACKED) != 0,
                                                                   (layout & WM_GEMM_OUT_PACKED) != 0));
}

extern "C" int wm_op_gemm16_takes_packed(int M, int N, int K) { return gemm16_takes_v5(M, N, K) ? 1 : 0; }

extern "C" int wm_op_ln_stats16(const float* x_dev, float* stats_dev, void* x16_dev, int64_t rows, int C, int precision, void* stream) {
    if (!x_dev || !stats_dev || !x16_dev) return fail("wm_op_ln_stats16: null buffer");
    return launch_ln_stats16(nullptr, (hipStream_t)stream, precision, x_dev, stats_dev, x16_dev, rows, C);
}

extern "C" int wm_op_ln_stats16_split(const float* x_dev, float* stats_dev, void* hi_dev, void* lo_dev, float* x_rw_dev, int64_t rows, int C,
                                      int precision, void* stream) {
    if (!x_dev || !stats_dev || !hi_dev || !lo_dev) return fail("wm_op_ln_stats16_split: null buffer");
    return launch_ln_stats16(nullptr, (hipStream_t)stream, precision, x_dev, stats_dev, hi_dev, rows, C, lo_dev, x_rw_dev, nullptr);
}

extern "C" int wm_op_stream_merge(const void* hi_dev, const void* lo_dev, float* out_dev, int64_t rows, int C, int precision, void* stream) {
    if (!hi_dev || !lo_dev || !out_dev) return fail("wm_op_stream_merge: null buffer");
    return launch_stream_merge(nullptr, (hipStream_t)stream, precision, hi_dev, lo_dev, out_dev, rows, C);
}

extern "C" int wm_op_gemm16_split(const void* a_dev, const void* w_dev, const float* bias_dev, void* hi_dev, void* lo_dev, float* stats_dev,
                                  int M, int N, int K, int layout, int precision, void* stream) {
    if (!a_dev || !w_dev || !hi_dev || !lo_dev || !stats_dev) return fail("wm_op_gemm16_split: null buffer");
    if (!gemm16_takes_v5(M, N, K)) return fail("wm_op_gemm16_split: M=%d N=%d K=%d is not served by the 256-row-tile kernel", M, N, K);
    GemmExtra x = GX((layout & WM_GEMM_W_PACKED) ? w_dev : nullptr, (layout & WM_GEMM_A_PACKED) != 0, 0);
    x.st_stats = stats_dev; x.res_hi = hi_dev; x.res_lo = lo_dev; x.out_lo = lo_dev;
    return launch_gemm16(nullptr, (hipStream_t)stream, precision, a_dev, (layout & WM_GEMM_W_PACKED) ? nullptr : w_dev, bias_dev, nullptr, 0,
                         nullptr, hi_dev, M, N, K, ACT_NONE, x);
}

extern "C" int wm_op_unpack16(const void* in_dev, void* out_dev, int64_t rows, int K, void* stream) {
    if (!in_dev || !out_dev || rows <= 0 || K <= 0 || rows % 16 || K % 32) return fail("wm_op_unpack16: rows=%lld K=%d (rows %% 16, K %% 32)", (long long)rows, K);
    hipLaunchKernelGGL(unpack16_lds_image_kernel, dim3(grid_for(rows * (K / 8))), dim3(256), 0, (hipStream_t)stream, (const uint4*)in_dev, (uint4*)out_dev, rows, K);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int wm_op_fold_weight16(const void* w16_dev, const float* gamma_dev, const float* beta_dev, const float* bias_dev, void* wf_dev,
                                   float* c1_dev, float* c2_dev, int N, int K, int precision, void* stream) {
    if (!w16_dev || !gamma_dev || !beta_dev || !wf_dev || !c1_dev || !c2_dev) return fail("wm_op_fold_weight16: null buffer");
    if (N <= 0 || K <= 0 || N % 16 || K % 32) return fail("wm_op_fold_weight16: N=%d K=%d (N %% 16, K %% 32)", N, K);
    if (precision == WM_PREC_FP16)
        hipLaunchKernelGGL(fold_weight_kernel<FP16>, dim3(N), dim3(256), 0, (hipStream_t)stream, (const u16*)w16_dev, (const float*)nullptr, gamma_dev, beta_dev, bias_dev, (u16*)wf_dev, c1_dev, c2_dev, N, K);
    else if (precision == WM_PREC_BF16)
        hipLaunchKernelGGL(fold_weight_kernel<BF16>, dim3(N), dim3(256), 0, (hipStream_t)stream, (const u16*)w16_dev, (const float*)nullptr, gamma_dev, beta_dev, bias_dev, (u16*)wf_dev, c1_dev, c2_dev, N, K);
    else return fail("wm_op_fold_weight16: precision %d", precision);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int wm_op_gemm16_folded(const void* x16_dev, const void* wf_dev, const float* c1_dev, const float* c2_dev, const float* stats_dev,
                                   float eps, void* out_16_dev, int M, int N, int K, int act, int precision, void* stream) {
    if (!x16_dev || !wf_dev || !c1_dev || !c2_dev || !stats_dev || !out_16_dev) return fail("wm_op_gemm16_folded: null buffer");
    const int out_packed = (act & WM_GEMM_OUT_PACKED) != 0;
    act &= ~WM_GEMM_OUT_PACKED;
    if (!gemm16_takes_v5(M, N, K)) return fail("wm_op_gemm16_folded: M=%d N=%d K=%d is not served by the 256-row-tile kernel", M, N, K);
    GemmExtra x = GX(wf_dev, 1, out_packed);
    x.fold_stats = stats_dev; x.fold_c1 = c1_dev; x.fold_eps = eps;
    return launch_gemm16(nullptr, (hipStream_t)stream, precision, x16_dev, nullptr, c2_dev, nullptr, 0, nullptr, out_16_dev, M, N, K, act, x);
}

extern "C" int wm_op_gemm16_stats(const void* a_dev, const void* w_dev, const float* bias_dev, const float* residual_dev, float* out_f32_dev,
                                  void* x16_dev, float* stats_dev, int M, int N, int K, int layout, int precision, void* stream) {
    if (!a_dev || !w_dev || !residual_dev || !out_f32_dev || !x16_dev || !stats_dev) return fail("wm_op_gemm16_stats: null buffer");
    if (!gemm16_takes_v5(M, N, K)) return fail("wm_op_gemm16_stats: M=%d N=%d K=%d is not served by the 256-row-tile kernel", M, N, K);
    GemmExtra x = GX((layout & WM_GEMM_W_PACKED) ? w_dev : nullptr, (layout & WM_GEMM_A_PACKED) != 0, 0);
    x.st_stats = stats_dev;
    return launch_gemm16(nullptr, (hipStream_t)stream, precision, a_dev, (layout & WM_GEMM_W_PACKED) ? nullptr : w_dev, bias_dev, residual_dev, 0,
                         out_f32_dev, x16_dev, M, N, K, ACT_NONE, x);
}

extern "C" int wm_op_pack16(const void* in_dev, void* out_dev, int64_t rows, int K, void* stream) {
    if (!in_dev || !out_dev || rows <= 0 || K <= 0 || rows % 16 || K % 32) return fail("wm_op_pack16: rows=%lld K=%d (rows %% 16, K %% 32)", (long long)rows, K);
    hipLaunchKernelGGL(pack16_lds_image_kernel, dim3(grid_for(rows * (K / 8))), dim3(256), 0, (hipStream_t)stream, (const uint4*)in_dev, (uint4*)out_dev, rows, K);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int wm_op_gemm8(const void* a_dev, const void* w_dev, const float* wscale_dev, const float* bias_dev, const float* residual_dev,
                           float* out_f32_dev, void* out_16_dev, void* out_8_dev, int M, int N, int K, int act, int precision, void* stream) {
    return launch_gemm8(nullptr, (hipStream_t)stream, precision, a_dev, w_dev, wscale_dev, bias_dev, residual_dev, out_f32_dev, out_16_dev, out_8_dev,
                        M, N, K, act);
}

extern "C" int wm_op_gemm8_planes(const void* a_dev, const void* w_dev, const float* wscale_dev, const float* bias_dev, void* hi_dev, void* lo_dev,
                                  int M, int N, int K, int precision, void* stream) {
    if (!hi_dev || !lo_dev) return fail("wm_op_gemm8_planes: null plane");
    return launch_gemm8(nullptr, (hipStream_t)stream, precision, a_dev, w_dev, wscale_dev, bias_dev, nullptr, nullptr, nullptr, nullptr, M, N, K, ACT_NONE,
                        hi_dev, lo_dev);
}

extern "C" int wm_op_stream_rows(float* x_f32_dev, void* hi_dev, void* lo_dev, int64_t rows, int C, int precision, int merge, void* stream) {
    if (!x_f32_dev || !hi_dev || !lo_dev) return fail("wm_op_stream_rows: null buffer");
    return launch_stream_rows(nullptr, (hipStream_t)stream, precision, x_f32_dev, hi_dev, lo_dev, rows, C, merge != 0);
}

extern "C" int wm_op_layernorm_fp8_plane(const void* hi_dev, const float* gamma_dev, const float* beta_dev, float eps, void* out_8_dev, int64_t rows, int C,
                                        int precision, void* stream) {
    if (!hi_dev || !out_8_dev || !gamma_dev || !beta_dev) return fail("wm_op_layernorm_fp8_plane: null buffer");
    return launch_layernorm_plane8(nullptr, (hipStream_t)stream, precision, hi_dev, gamma_dev, beta_dev, eps, out_8_dev, rows, C);
}

extern "C" int wm_op_cvt_f32_to_fp8(const float* in_dev, void* out_dev, int64_t n, void* stream) {
    if (n % 4) return fail("cvt fp8: n must be a multiple of 4");
    hipLaunchKernelGGL(cvt_f32_to_fp8_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, in_dev, (unsigned char*)out_dev, n / 4);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int wm_op_conv3x3_16(const void* a_dev, const void* w_dev, float* out_dev, int batch, int c_out, int c_in, int precision,
                               void* stream) {
    return launch_conv3x3_16(nullptr, (hipStream_t)stream, precision, a_dev, w_dev, out_dev, batch * 4096, c_out, c_in);
}

extern "C" int wm_op_patch_embed16(const void* img16_dev, const void* w_dev, const float* bias_dev, float* out_f32_dev, void* out_16_dev,
                                  int batch, int n_out, int c_in, int precision, void* stream) {
    if (!img16_dev || !w_dev) return fail("wm_op_patch_embed16: null buffer");
    return launch_patch_embed16(nullptr, (hipStream_t)stream, precision, img16_dev, w_dev, bias_dev, nullptr, 0, out_f32_dev, out_16_dev, batch, n_out, c_in);
}

extern "C" int wm_op_gemm32(const float* a_dev, const float* w_dev, const float* bias_dev, const float* residual_dev, float* out_dev,
                            int M, int N, int K, int act, void* stream) {
    // act & WM_GEMM32_SPLIT: the fp16-split form (gemm32x3_kernel, W split per K-step); otherwise the fp32-MFMA kernel
    const int split = (act & WM_GEMM32_SPLIT) != 0;
    return launch_gemm32(nullptr, (hipStream_t)stream, a_dev, w_dev, bias_dev, residual_dev, out_dev, M, N, K, act & 0xff, 0, split ? 2 : 1);
}

extern "C" int wm_op_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps, float* out_f32_dev,
                               void* out_16_dev, int64_t rows, int C, int precision, void* stream) {
    const int packed = (precision & WM_LAYOUT_PACKED) != 0;
    precision &= ~WM_LAYOUT_PACKED;
    if (!out_f32_dev && out_16_dev)      // the transformer blocks' form: column-tiled statistics (the arithmetic of the folded LayerNorm's statistics)
        return launch_layernorm_block(nullptr, (hipStream_t)stream, precision, x_dev, gamma_dev, beta_dev, eps, out_16_dev, rows, C, packed);
    if (packed) return fail("wm_op_layernorm: the LDS-image-order output exists for the 16-bit-only form");
    return launch_layernorm(nullptr, (hipStream_t)stream, precision, x_dev, gamma_dev, beta_dev, eps, out_f32_dev, out_16_dev, rows, C);
}

extern "C" int wm_op_encoder_attention(const void* qkv_dev, const float* qkv_bias_dev, const float* rel_pos_h_dev,
                                       const float* rel_pos_w_dev, void* out_dev, int batch, int heads, int head_dim, int window,
                                       int precision, void* stream) {
    return launch_encoder_attention(nullptr, (hipStream_t)stream, precision, qkv_dev, qkv_bias_dev, rel_pos_h_dev, rel_pos_w_dev, out_dev,
                                    batch, heads, head_dim, window);
}

extern "C" int wm_op_encoder_attention_qkv(const void* q_dev, const void* k_dev, const void* v_dev, int token_stride, const float* qkv_bias_dev,
                                           const float* rel_pos_h_dev, const float* rel_pos_w_dev, void* out_dev, int batch, int heads,
                                           int head_dim, int window, int precision, void* stream) {
    return launch_encoder_attention(nullptr, (hipStream_t)stream, precision, q_dev, qkv_bias_dev, rel_pos_h_dev, rel_pos_w_dev, out_dev,
                                    batch, heads, head_dim, window, nullptr, k_dev, v_dev, token_stride);
}

extern "C" int wm_op_mha16(const void* q_dev, int q_stride, const void* k_dev, int k_stride, const void* v_dev, int v_stride,
                           void* out_dev, int out_stride, int batch, int heads, int head_dim, int nq, int nk, int precision, void* stream) {
    return launch_mha16(nullptr, (hipStream_t)stream, precision, q_dev, q_stride, k_dev, k_stride, v_dev, v_stride, out_dev, out_stride,
                        batch, heads, head_dim, nq, nk);
}

extern "C" int wm_op_mha32(const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, int batch, int heads,
                           int head_dim, int nq, int nk, void* stream) {
    return launch_mha32(nullptr, (hipStream_t)stream, q_dev, k_dev, v_dev, out_dev, batch, heads, head_dim, nq, nk);
}
