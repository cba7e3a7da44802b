// C ABI of libgptqhip.so (declared in include/gptqhip.h): argument validation, workspace carving,
// kernel selection.  No torch types and no process-global mutable state: the error string and the tuning
// overrides (benchmarks / tests) are THREAD-LOCAL, so concurrent callers on different threads, devices or
// streams never observe each other's settings.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/gptqhip.h"
#include "gptqhip_codes.h"
#include "gptqhip_device.h"
#include "gptqhip_host.h"

namespace gptqhip {

static thread_local char g_err[512] = "";
// per-thread overrides (gptqhip_set_tuning); 0 = not overridden by this thread
static thread_local int t_force_split = 0;
static thread_local int t_force_kernel = 0;
static thread_local int t_force_waves = 0;
static thread_local int t_decode_form = -1;   // gptqhip_set_decode_form: -1 = the process default

// process-wide DEFAULTS from the environment, read once and immutable afterwards (triage switches, like the env flags
// the reference steers its kernels with, torch.py:172-190):
//   GPTQHIP_FORCE_KERNEL=1|2  always the decode (skinny) / prefill (tiled) kernel;  GPTQHIP_FORCE_SPLIT_K=n;
//   GPTQHIP_FORCE_VARIANT=n   decode: waves per block; prefill: 1/2/3 = 256/128/64-row tiles
struct EnvTuning {
    int split, kernel, waves;
};
static int env_int(const char* name) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : 0;
}
static const EnvTuning& env_tuning() {
    static const EnvTuning e = {env_int("GPTQHIP_FORCE_SPLIT_K"), env_int("GPTQHIP_FORCE_KERNEL"), env_int("GPTQHIP_FORCE_VARIANT")};
    return e;
}
// batch-1 decode form (include/gptqhip.h gptqhip_set_decode_form).  Process default: 5 (preload + raw codes as fp16 denormals) for fp16 activations
// with fp16 scales and for bf16 activations (converted to fp16 per wave, exactly); 4 (preload, the reference's per-weight rounding) for fp16 activations
// with bf16 scales, and for everything when GPTQHIP_DECODE_BITFAITHFUL=1 is in the environment.
static int decode_form_for(int act_dtype, int scale_dtype) {
    static const int bitfaithful = env_int("GPTQHIP_DECODE_BITFAITHFUL");
    const bool exact_ok = act_dtype == GPTQHIP_FP16 && scale_dtype == GPTQHIP_FP16;   // the exact-arithmetic forms exist for fp16 x fp16 only
    const bool raw_ok = exact_ok || act_dtype == GPTQHIP_BF16;                        // ... the raw-code form also for bf16 activations (any scale dtype)
    if (t_decode_form >= 0) {
        if (t_decode_form == 5) return raw_ok ? 5 : 4;
        return (t_decode_form == 1 && !exact_ok) ? 4 : t_decode_form;
    }
    return (bitfaithful || !raw_ok) ? 4 : 5;
}
#define g_force_split (t_force_split ? t_force_split : env_tuning().split)
#define g_force_kernel (t_force_kernel ? t_force_kernel : env_tuning().kernel)
#define g_force_waves (t_force_waves ? t_force_waves : env_tuning().waves)
// (values >= 32 of the variant override name a tile height of the prefill kernel, not a wave count of the decode kernel)
#define g_skinny_waves (g_force_waves >= 32 ? 0 : g_force_waves)
// (1064 / 1128: 128-column blocks of the prefill kernel)

constexpr size_t kInKernelPermMaxRowBytes = 44 * 1024;   // AM_ROW1P keeps the x row in LDS next to 16 KiB of per-wave slots

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return GPTQHIP_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return GPTQHIP_EHIP;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int kSkinnyMaxM = 32;  // above: the MFMA-tiled kernel (split-K when its grid is small)
constexpr int kSkinnyMaxRows4 = 64;  // rows one decode-kernel launch takes with 4-bit weights (MT = 4 instantiations)
constexpr size_t kCounterBytes = 64 * 1024;  // 16384 arrival counters

struct WorkspaceLayout {
    size_t counters_off, counters_bytes;
    size_t gather_off, gather_bytes;
    size_t slabs_off, slabs_bytes;
    size_t total;
};

static WorkspaceLayout layout_workspace(int M, int K, int N, int group_size, int bits, int has_perm) {
    WorkspaceLayout L;
    L.counters_off = 0;
    // FIXED-size counter region: it must stay zero between calls, so no other data may ever alias it
    // whatever N the previous call had (the kernels reset their counters after use).
    L.counters_bytes = kCounterBytes;
    L.gather_off = L.counters_off + L.counters_bytes;
    L.gather_bytes = has_perm ? align_up((size_t)M * K * 2, 256) : 0;
    L.slabs_off = L.gather_off + L.gather_bytes;
    const int mchunk = M < kSkinnyMaxM ? M : kSkinnyMaxM;
    const int mchunk4 = M < kSkinnyMaxRows4 ? M : kSkinnyMaxRows4;   // (4-bit: up to 64 rows per launch, planned without cross-block split)
    // worst case over the split heuristics: the skinny plan for one row-chunk
    // the SAME plans gptqhip_gemm will make for this (shape, group_size, bits): both sides call these planners with
    // identical arguments, so the layout cannot drift from the launch
    size_t floats = plan_skinny(mchunk, K, N, group_size, g_force_split, g_skinny_waves, false, bits).slab_floats;
    if (mchunk == 1) {   // the preload form of the batch-1 kernel plans without 2-deep rings (plan_skinny prefer_deep): cover both
        for (int deep = 1; deep <= 2; ++deep) {
            const size_t f1 = plan_skinny(1, K, N, group_size, g_force_split, g_skinny_waves, false, bits, 0, deep).slab_floats;
            if (f1 > floats) floats = f1;
        }
    }
    {
        const size_t f4 = plan_skinny(mchunk4, K, N, group_size, g_force_split, g_skinny_waves, false, bits).slab_floats;
        if (f4 > floats) floats = f4;
    }
    if (M > 16 || g_force_kernel == 2) {
        const TiledPlan tp = plan_tiled(M, K, N, group_size, bits, g_force_waves, g_force_split);
        if (tp.slab_floats > floats) floats = tp.slab_floats;
    }
    L.slabs_bytes = align_up(floats * sizeof(float), 256);
    L.total = L.slabs_off + L.slabs_bytes;
    return L;
}

// Kernel family of one gptqhip_gemm call (the measured crossover; the comments at its use in gptqhip_gemm give the numbers)
static bool gemm_uses_tiled(int M, int K, int N, int group_size, int bits) {
    bool wide = N >= 8192 && M > 16;
    if (wide && M <= 32 && N < 65536 && g_force_kernel == 0) {
        // 17..32 rows on a wide layer: the decode kernel's wide form where it is plannable and either K is short or its blocks fit ONE
        // round of the chip (8192x8192: 256 blocks, 13.0-13.5 us vs 16.4-17.4 tiled; 8192x10240: 320 blocks, 19.1-22.3 vs 18.9-19.9)
        const int nt = plan_skinny(M, K, N, group_size, g_force_split, g_skinny_waves, false, bits, 1).nt;
        if (nt > 1 && (K < 8192 || ceil_div(ceil_div(N, kTileN), nt) <= 256)) wide = false;
    }
    // 33..64 rows in one launch of the decode kernel: 4-bit, short K, narrow layers (4096^2: 8.8-10.7 us vs 12.2-14.8 tiled; at
    // N = 6144 the prefill kernel is level or ahead since its round-3 retuning: 13.7-17.6 vs 15.2-18.0 us; profiles/r03_mid_m_sweep.txt)
    int skinny_max = (bits == 4 && K < 8192 && N < 6144) ? kSkinnyMaxRows4 : kSkinnyMaxM;
    // round 5 (128-column blocks + the refitted planner): on K-heavy layers the prefill kernel leads from 17 rows -- 14336x4096 13.0 vs
    // 15.8 us at M = 17, 13.1 vs 18.1 at 32; 11008x4096 11.8 vs 12.8 / 12.6 vs 15.1; 28672x8192 35.0 vs 39.5 / 33.9 vs 41.2 -- while 8192x8192
    // and 8192x1024 stay with the decode kernel up to 32 rows (13.1-13.5 vs 13.7-14.1; 9.2-10.6 vs 10.4-10.9)
    if (bits == 4 && group_size % kChunkK == 0 && K >= 10240) skinny_max = 16;
    return (g_force_kernel == 2) || (g_force_kernel == 0 && (M > skinny_max || wide));
}

}  // namespace gptqhip

using namespace gptqhip;

extern "C" {

int gptqhip_abi_version(void) { return GPTQHIP_ABI_VERSION; }

const char* gptqhip_last_error(void) { return g_err; }

int gptqhip_set_tuning(int force_split_k, int force_kernel, int force_waves) {
    t_force_split = force_split_k;
    t_force_kernel = force_kernel;
    t_force_waves = force_waves;
    return GPTQHIP_OK;
}

int gptqhip_set_decode_form(int form) {
    if (form < -1 || form > 5) {
        set_error("gptqhip_set_decode_form: form=%d (0..5, -1 = process default; see include/gptqhip.h)", form);
        return GPTQHIP_EINVAL;
    }
    t_decode_form = form;
    return GPTQHIP_OK;
}

int gptqhip_device_info(int device, int* cu_count, size_t* hbm_bytes, char* arch, int arch_len) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        set_error("gptqhip_device_info: no HIP device %d (count %d)", device, n);
        return GPTQHIP_ENODEV;
    }
    hipDeviceProp_t prop;
    int rc = check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (rc) return rc;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("gptqhip: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
        return GPTQHIP_ENODEV;
    }
    return GPTQHIP_OK;
}

size_t gptqhip_workspace_bytes(int M, int K, int N, int group_size, int bits, int has_perm) {
    if (M <= 0 || K <= 0 || N <= 0 || group_size <= 0 || K % group_size != 0 || (bits != 4 && bits != 8)) return 0;
    return layout_workspace(M, K, N, group_size, bits, has_perm).total;
}

static int validate_common(const char* fn, int K, int N, int group_size, int bits) {
    if (bits != 4 && bits != 8) {
        set_error("%s: bits=%d unsupported (4 or 8)", fn, bits);
        return GPTQHIP_EINVAL;
    }
    if (K <= 0 || N <= 0 || K % 32 != 0 || N % 8 != 0) {
        set_error("%s: K=%d must be a positive multiple of 32 and N=%d of 8", fn, K, N);
        return GPTQHIP_EINVAL;
    }
    if (group_size <= 0 || group_size % 32 != 0 || K % group_size != 0) {
        set_error("%s: group_size=%d must be a multiple of 32 dividing K=%d", fn, group_size, K);
        return GPTQHIP_EINVAL;
    }
    return GPTQHIP_OK;
}

size_t gptqhip_tiled_words(int K, int N, int bits) {
    if (K <= 0 || N <= 0 || (bits != 4 && bits != 8)) return 0;
    return (size_t)ceil_div(N, kTileN) * ceil_div(K, kChunkK) * (bits == 4 ? 256 : 512);
}

size_t gptqhip_meta_words(int K, int N, int group_size) {
    if (K <= 0 || N <= 0 || group_size <= 0 || K % group_size != 0) return 0;
    return (size_t)ceil_div(N, kTileN) * (K / group_size) * 16;
}

int gptqhip_repack_tiled(const int32_t* qweight, const int32_t* qzeros, const void* scales, const int32_t* perm,
                         uint32_t* qweight_t, uint32_t* meta, int K, int N, int group_size, int bits,
                         gptqhip_stream_t stream) {
    if (!qzeros || !scales || !meta || ((qweight == nullptr) != (qweight_t == nullptr))) {
        set_error("gptqhip_repack_tiled: null tensor pointer (qweight/qweight_t may only be NULL together)");
        return GPTQHIP_EINVAL;
    }
    int rc = validate_common("gptqhip_repack_tiled", K, N, group_size, bits);
    if (rc) return rc;
    return launch_repack_tiled(qweight, qzeros, scales, perm, qweight_t, meta, K, N, group_size, bits,
                               reinterpret_cast<hipStream_t>(stream));
}

int gptqhip_gemm(const void* x, const uint32_t* qweight, const uint32_t* meta,
                 const int32_t* perm, const void* bias, void* out, void* workspace, size_t workspace_bytes, int M,
                 int K, int N, int group_size, int bits, int act_dtype, int scale_dtype, int flags,
                 gptqhip_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (M == 0) return GPTQHIP_OK;  // empty batch: nothing to do (reference returns an empty tensor)
    if (!x || !qweight || !meta || !out) {
        set_error("gptqhip_gemm: null tensor pointer");
        return GPTQHIP_EINVAL;
    }
    if (M < 0) {
        set_error("gptqhip_gemm: M=%d", M);
        return GPTQHIP_EINVAL;
    }
    int rc = validate_common("gptqhip_gemm", K, N, group_size, bits);
    if (rc) return rc;
    if ((act_dtype != GPTQHIP_FP16 && act_dtype != GPTQHIP_BF16) ||
        (scale_dtype != GPTQHIP_FP16 && scale_dtype != GPTQHIP_BF16)) {
        set_error("gptqhip_gemm: dtype tags must be GPTQHIP_FP16/BF16");
        return GPTQHIP_EINVAL;
    }
    if ((size_t)ceil_div(N, kTileN) * sizeof(int) > kCounterBytes) {
        set_error("gptqhip_gemm: N=%d too large (max %zu columns)", N, kCounterBytes / sizeof(int) * kTileN);
        return GPTQHIP_EINVAL;
    }
    const bool partial_f32 = (flags & GPTQHIP_GEMM_PARTIAL_F32) != 0;
    if ((flags & ~(GPTQHIP_GEMM_PARTIAL_F32 | GPTQHIP_GEMM_EXACT_BF16)) != 0 || (partial_f32 && bias != nullptr)) {
        set_error("gptqhip_gemm: bad flags 0x%x (PARTIAL_F32 excludes bias)", flags);
        return GPTQHIP_EINVAL;
    }
    const WorkspaceLayout L = layout_workspace(M, K, N, group_size, bits, perm != nullptr);
    if (!workspace || workspace_bytes < L.total) {
        set_error("gptqhip_gemm: workspace %zu bytes < required %zu", workspace_bytes, L.total);
        return GPTQHIP_ENOMEM;
    }
    char* ws = reinterpret_cast<char*>(workspace);
    int* counters = reinterpret_cast<int*>(ws + L.counters_off);
    float* slabs = reinterpret_cast<float*>(ws + L.slabs_off);

    const void* xin = x;
    // act-order: batch-1 decode applies the permutation inside the kernel (no extra launch per linear) when the plan is
    // the regular straight-line pipeline; everything else gathers x once into the workspace first
    bool fused_perm = false;
    if (perm && M == 1 && g_force_kernel != 2) {
        const SkinnyPlan pl1 = plan_skinny(1, K, N, group_size, g_force_split, g_skinny_waves, true);
        // (the kernel keeps the x row in LDS next to 16 KiB of per-wave slots: stay inside the default 64 KiB of dynamic LDS)
        fused_perm = pl1.regular && pl1.gpc == 1 && pl1.depth == 4 && (size_t)K * 2 <= kInKernelPermMaxRowBytes;
    }
    if (perm && !fused_perm) {
        void* gbuf = ws + L.gather_off;
        rc = launch_gather_cols(x, perm, gbuf, M, K, stream);
        if (rc) return rc;
        xin = gbuf;
    }

    const int decode_form = decode_form_for(act_dtype, scale_dtype);
    GemmArgs a;
    a.qweight = qweight;
    a.meta = meta;
    a.bias = bias;
    a.K = K;
    a.N = N;
    a.group_size = group_size;
    a.bits = bits;
    a.act_dtype = act_dtype;
    a.scale_dtype = scale_dtype;
    a.out_f32 = partial_f32 ? 1 : 0;
    a.perm = fused_perm ? perm : nullptr;
    a.exact_bf16 = ((flags & GPTQHIP_GEMM_EXACT_BF16) && act_dtype == GPTQHIP_BF16) ? 1 : 0;   // (no fp16 form: the flag changes nothing there)
    a.alg_fp16 = decode_form == 5 ? 2 : ((decode_form == 2 || decode_form == 3) ? 1 : 0);
    a.preload = (decode_form == 3 || decode_form == 4 || decode_form == 5) ? 1 : 0;

    // measured crossover (profiles/r03_mid_m_sweep.txt, round 3: split-ring pipeline for 17..64 rows, 33..64 rows in one launch with
    // 4-bit weights): the decode kernel leads up to 64 rows on layers with K < 8192 and N < 6144 (4096^2 at M=64 10.7 us vs 14.8 us
    // tiled; at N = 6144 the retuned prefill kernel is level or ahead: 17.6 vs 18.0), up to 32 rows on long-K layers (14336x4096: 18.4 vs 18.1 us at M=32, 24.6 vs 19.2 at M=40), and
    // up to 16 rows on wide layers (N >= 8192, e.g. fused gate_up: 28.4 vs 25.6 us at M=24) where every 16-column block of the decode
    // kernel re-stages the whole activation tile.  Everything above goes to the MFMA-tiled kernel (64-row tiles, split-K).
    // wide layers (N >= 8192): up to 16 rows the decode kernel (its wide-layer form where plannable: several column tiles per block
    // share one staging of the activation tile -- 4096x28672 at M=16 23.7 -> 16.7 us, 8192x57344 80.5 -> 53.2, lm_head 87.4 -> 55.6);
    // 17..32 rows too when N < 65536 and K < 8192 or the blocks fit one round (4096x28672 at M=32 32.4 -> 24.4 us vs 27.3 us tiled,
    // 4096x8192 11.9 -> 8.7 vs 13.0, 8192x8192 13.5 vs 17.4; but 8192x10240 22.3 vs 19.9 tiled, 4096x128256 91.9 vs 80.7); above that the tiled kernel (4096x28672 at M=48: 28.7 vs 39.0 us).
    // profiles/r03_wide_layers.txt
    const bool use_tiled = gemm_uses_tiled(M, K, N, group_size, bits);
    // rows per decode-kernel launch: 32, or 64 with 4-bit weights (one launch, the weights are streamed once)
    const int rows_per_launch = bits == 4 ? kSkinnyMaxRows4 : kSkinnyMaxM;
    if (use_tiled) {
        a.x = xin;
        a.out = out;
        a.M = M;
        const TiledPlan tp = plan_tiled(M, K, N, group_size, bits, g_force_waves, g_force_split);  // force_waves doubles as tiled variant
        if (tp.tail_cols > 0) {
            // two launches (plan_tiled): 256-row tiles over the leading block columns, 128-row tiles over the trailing
            // ones.  The tile-major weight layout makes a column sub-range a plain pointer offset; the output keeps its
            // row stride (ldo).
            const int n_a = (ceil_div(N, kTiledBN) - tp.tail_cols) * kTiledBN;
            GemmArgs a1 = a;
            a1.N = n_a;
            a1.ldo = N;
            TiledPlan t1 = tp;
            t1.tail_cols = 0;
            rc = launch_tiled(a1, t1, slabs, stream);
            if (rc) return rc;
            const size_t tile_off = (size_t)n_a / kTileN;
            const size_t chunks = (size_t)ceil_div(K, kChunkK);
            GemmArgs a2 = a;
            a2.N = N - n_a;
            a2.ldo = N;
            a2.qweight = a.qweight + tile_off * chunks * (bits == 4 ? 1 : 2) * 256;
            a2.meta = a.meta + tile_off * (size_t)(K / group_size) * 16;
            a2.bias = a.bias ? reinterpret_cast<const char*>(a.bias) + (size_t)n_a * 2 : nullptr;
            a2.out = reinterpret_cast<char*>(out) + (size_t)n_a * (partial_f32 ? 4 : 2);
            TiledPlan t2 = t1;
            t2.bm = 128;
            return launch_tiled(a2, t2, slabs, stream);
        }
        return launch_tiled(a, tp, slabs, stream);
    }
    // batch 1: the stream form (LDS-DMA ring + algebraic dequant, gptqhip_stream.hip) unless the caller asked for the bit-faithful chain
    if (M == 1 && bits == 4 && !fused_perm && decode_form == 1 && g_force_kernel == 0 && g_force_split == 0 && !a.exact_bf16) {
        const StreamPlan sp = plan_stream(K, N, group_size, bits, g_skinny_waves, false, bias != nullptr);
        if (sp.ok) {
            a.x = xin;
            a.out = out;
            a.M = 1;
            return launch_stream(a, sp, stream);
        }
    }
    // skinny kernel, kSkinnyMaxM rows per launch
    for (int m0 = 0; m0 < M; m0 += rows_per_launch) {
        const int mc = (M - m0) < rows_per_launch ? (M - m0) : rows_per_launch;
        a.x = reinterpret_cast<const char*>(xin) + (size_t)m0 * K * 2;
        a.out = reinterpret_cast<char*>(out) + (size_t)m0 * N * (partial_f32 ? 4 : 2);
        a.M = mc;
        const SkinnyPlan pl = plan_skinny(mc, K, N, group_size, g_force_split, g_skinny_waves, fused_perm, bits, 1, (a.preload && mc == 1 && !fused_perm) ? (a.alg_fp16 == 2 ? 2 : 1) : 0);
        rc = launch_skinny(a, pl, slabs, counters, stream);
        if (rc) return rc;
    }
    return GPTQHIP_OK;
}

int gptqhip_plan_describe(int M, int K, int N, int group_size, int bits, int has_perm, char* buf, int buf_len) {
    if (!buf || buf_len <= 0) {
        set_error("gptqhip_plan_describe: no buffer");
        return GPTQHIP_EINVAL;
    }
    int rc = validate_common("gptqhip_plan_describe", K, N, group_size, bits);
    if (rc) return rc;
    if (M <= 0) {
        set_error("gptqhip_plan_describe: M=%d", M);
        return GPTQHIP_EINVAL;
    }
    if (gemm_uses_tiled(M, K, N, group_size, bits)) {
        const TiledPlan tp = plan_tiled(M, K, N, group_size, bits, g_force_waves, g_force_split);
        if (tp.bn != 256)
            snprintf(buf, (size_t)buf_len, "tiled bm=%d bn=%d splits=%d tail_cols=%d gather=%d", tp.bm, tp.bn, tp.splits, tp.tail_cols, has_perm ? 1 : 0);
        else
            snprintf(buf, (size_t)buf_len, "tiled bm=%d splits=%d tail_cols=%d gather=%d", tp.bm, tp.splits, tp.tail_cols, has_perm ? 1 : 0);
        return GPTQHIP_OK;
    }
    const int rows = bits == 4 ? kSkinnyMaxRows4 : kSkinnyMaxM;
    const int mc = M < rows ? M : rows;
    bool fused_perm = false;
    if (has_perm && M == 1) {
        const SkinnyPlan pl1 = plan_skinny(1, K, N, group_size, g_force_split, g_skinny_waves, true);
        fused_perm = pl1.regular && pl1.gpc == 1 && pl1.depth == 4 && (size_t)K * 2 <= kInKernelPermMaxRowBytes;
    }
    // batch 1, 4-bit: the geometry of the process-default decode form for fp16 activations (the preload forms plan without 2-deep rings, the
    // raw-code form runs q|k|v-shaped layers with four waves: plan_skinny's prefer_deep)
    int deep = 0;
    if (mc == 1 && bits == 4 && !fused_perm) {
        const int form = decode_form_for(GPTQHIP_FP16, GPTQHIP_FP16);
        deep = form == 5 ? 2 : ((form == 3 || form == 4) ? 1 : 0);
    }
    const SkinnyPlan pl = plan_skinny(mc, K, N, group_size, g_force_split, g_skinny_waves, fused_perm, bits, 1, deep);
    snprintf(buf, (size_t)buf_len, "skinny launches=%d mt=%d nt=%d waves=%d depth=%d regular=%d splits=%d gather=%d", ceil_div(M, rows), pl.mt,
             pl.nt, pl.waves, pl.depth, pl.regular, pl.splits, has_perm && !fused_perm ? 1 : 0);
    return GPTQHIP_OK;
}

int gptqhip_decode_supported(int K, int N, int group_size, int has_perm, int M) {
    // the decode op rides on the skinny kernel's regular batch-1 pipeline (straight-line counted-wait ring, one group
    // constant per 128-row chunk): same predicate as the planner's
    if (K <= 0 || N <= 0 || group_size <= 0 || K % group_size != 0 || K % 32 != 0 || N % 8 != 0) return 0;
    if (M < 1 || M > 16 || (M > 1 && has_perm)) return 0;
    const SkinnyPlan pl = plan_skinny(M, K, N, group_size, 0, 0, has_perm != 0);
    if (has_perm && !(pl.depth == 4 && (size_t)K * 2 <= kInKernelPermMaxRowBytes)) return 0;
    // (group_size 32 / 64 -- a group constant per K-step -- rides the same pipeline; the in-kernel permutation needs one per chunk)
    return (pl.regular && (pl.gpc == 1 || !has_perm) && pl.mt == 1) ? 1 : 0;
}

int gptqhip_decode_linear(const gptqhip_decode_op* op, gptqhip_stream_t stream) {
    if (!op || !op->qweight_t || !op->meta || !op->x || !op->out) {
        set_error("gptqhip_decode_linear: null tensor pointer");
        return GPTQHIP_EINVAL;
    }
    int rc = validate_common("gptqhip_decode_linear", op->K, op->N, op->group_size, op->bits);
    if (rc) return rc;
    if ((op->act_dtype != GPTQHIP_FP16 && op->act_dtype != GPTQHIP_BF16) ||
        (op->scale_dtype != GPTQHIP_FP16 && op->scale_dtype != GPTQHIP_BF16)) {
        set_error("gptqhip_decode_linear: dtype tags must be GPTQHIP_FP16/BF16");
        return GPTQHIP_EINVAL;
    }
    if (op->in_glue < GPTQHIP_GLUE_NONE || op->in_glue > GPTQHIP_GLUE_SILU_MUL ||
        (op->in_glue == GPTQHIP_GLUE_RMSNORM && !op->norm_weight)) {
        set_error("gptqhip_decode_linear: bad in_glue %d (RMSNORM needs norm_weight)", op->in_glue);
        return GPTQHIP_EINVAL;
    }
    if ((size_t)ceil_div(op->N, kTileN) * sizeof(int) > kCounterBytes) {
        set_error("gptqhip_decode_linear: N=%d too large", op->N);
        return GPTQHIP_EINVAL;
    }
    if (op->out_glue == GPTQHIP_OUT_PARTIAL_F32 && (op->bias || op->residual || op->stats_out)) {
        set_error("gptqhip_decode_linear: OUT_PARTIAL_F32 excludes bias / residual / stats_out");
        return GPTQHIP_EINVAL;
    }
    if (op->out_glue != GPTQHIP_OUT_NONE && op->out_glue != GPTQHIP_OUT_SILU_MUL_PAIRED && op->out_glue != GPTQHIP_OUT_PARTIAL_F32) {
        set_error("gptqhip_decode_linear: bad out_glue %d", op->out_glue);
        return GPTQHIP_EINVAL;
    }
    if (op->out_glue == GPTQHIP_OUT_SILU_MUL_PAIRED && (op->N % 16 != 0 || op->residual || op->stats_out)) {
        set_error("gptqhip_decode_linear: OUT_SILU_MUL_PAIRED needs N %% 16 == 0 and excludes residual / stats_out");
        return GPTQHIP_EINVAL;
    }
    if (op->stats_in && (op->in_glue != GPTQHIP_GLUE_RMSNORM || op->stats_n <= 0 || op->stats_n > 512)) {
        set_error("gptqhip_decode_linear: stats_in needs in_glue RMSNORM and 1..512 partial sums (got %d)", op->stats_n);
        return GPTQHIP_EINVAL;
    }
    const int M = op->M;
    if (M < 1 || M > 16 || (M > 1 && (op->perm || op->in_glue == GPTQHIP_GLUE_SILU_MUL))) {
        set_error("gptqhip_decode_linear: M=%d outside 1..16, or M > 1 with perm / SiLU*mul input glue", M);
        return GPTQHIP_EINVAL;
    }
    const int decode_form = decode_form_for(op->act_dtype, op->scale_dtype);
    if (M == 1 && op->bits == 4 && !op->perm && op->in_glue != GPTQHIP_GLUE_SILU_MUL && decode_form == 1 && g_force_split == 0 &&
        !(op->flags & GPTQHIP_GEMM_EXACT_BF16)) {
        const StreamPlan sp = plan_stream(op->K, op->N, op->group_size, op->bits, g_skinny_waves, op->in_glue == GPTQHIP_GLUE_RMSNORM,
                                           op->bias != nullptr || op->residual != nullptr);
        if (sp.ok) {
            GemmArgs a;
            a.x = op->x;
            a.qweight = op->qweight_t;
            a.meta = op->meta;
            a.bias = op->bias;
            a.out = op->out;
            a.M = 1;
            a.K = op->K;
            a.N = op->N;
            a.group_size = op->group_size;
            a.bits = op->bits;
            a.act_dtype = op->act_dtype;
            a.scale_dtype = op->scale_dtype;
            a.out_f32 = op->out_glue == GPTQHIP_OUT_PARTIAL_F32 ? 1 : 0;
            a.in_glue = op->in_glue;
            a.glue_b = op->norm_weight;
            a.residual = op->residual;
            a.eps = op->eps;
            a.stats_in = op->stats_in;
            a.stats_n = op->stats_n;
            a.stats_out = op->stats_out;
            a.out_glue = op->out_glue == GPTQHIP_OUT_PARTIAL_F32 ? GPTQHIP_OUT_NONE : op->out_glue;
            return launch_stream(a, sp, reinterpret_cast<hipStream_t>(stream));
        }
    }
    // wide layers at 5..16 rows: the decode kernel's wide form serves the RMSNorm-in / paired-SiLU-out ops (gate_up); residual /
    // statistics epilogues and the SiLU*mul input glue stay with the one-tile kernel
    const bool wide_ok = M >= 2 && !op->perm && !op->residual && !op->stats_out && op->bits == 4 && op->group_size % kChunkK == 0 &&
                         (op->in_glue == GPTQHIP_GLUE_RMSNORM || (op->in_glue == GPTQHIP_GLUE_NONE && op->out_glue == GPTQHIP_OUT_NONE)) &&
                         (op->out_glue == GPTQHIP_OUT_NONE || op->out_glue == GPTQHIP_OUT_SILU_MUL_PAIRED ||
                          (op->out_glue == GPTQHIP_OUT_PARTIAL_F32 && op->in_glue == GPTQHIP_GLUE_NONE));
    const bool preload_form = (decode_form == 3 || decode_form == 4 || decode_form == 5) && M == 1 && !op->perm && op->bits == 4;
    const SkinnyPlan pl = plan_skinny(M, op->K, op->N, op->group_size, g_force_split, g_skinny_waves, op->perm != nullptr, op->bits,
                                      !wide_ok ? 0 : (op->in_glue == GPTQHIP_GLUE_RMSNORM ? 2 : 1), preload_form ? (decode_form == 5 ? 2 : 1) : 0);
    if (op->perm && !(pl.depth == 4 && (size_t)op->K * 2 <= kInKernelPermMaxRowBytes)) {
        set_error("gptqhip_decode_linear: K=%d is outside the in-kernel act-order variant (gather x and pass perm = NULL)", op->K);
        return GPTQHIP_EINVAL;
    }
    if (!(pl.regular && (pl.gpc == 1 || !op->perm) && pl.mt == 1)) {
        set_error("gptqhip_decode_linear: K=%d group_size=%d is outside the decode op's regular pipeline (use gptqhip_gemm)",
                  op->K, op->group_size);
        return GPTQHIP_EINVAL;
    }
    const WorkspaceLayout L = layout_workspace(M, op->K, op->N, op->group_size, op->bits, 0);
    if (pl.splits > 1 && (!op->workspace || op->workspace_bytes < L.total)) {
        set_error("gptqhip_decode_linear: workspace %zu bytes < required %zu", op->workspace_bytes, L.total);
        return GPTQHIP_ENOMEM;
    }
    char* ws = reinterpret_cast<char*>(op->workspace);
    GemmArgs a;
    a.x = op->x;
    a.exact_bf16 = ((op->flags & GPTQHIP_GEMM_EXACT_BF16) && op->act_dtype == GPTQHIP_BF16) ? 1 : 0;
    a.alg_fp16 = decode_form == 5 ? 2 : ((decode_form == 2 || decode_form == 3) ? 1 : 0);
    a.preload = (decode_form == 3 || decode_form == 4 || decode_form == 5) ? 1 : 0;
    a.perm = op->perm;
    a.qweight = op->qweight_t;
    a.meta = op->meta;
    a.bias = op->bias;
    a.out = op->out;
    a.M = M;
    a.K = op->K;
    a.N = op->N;
    a.group_size = op->group_size;
    a.bits = op->bits;
    a.act_dtype = op->act_dtype;
    a.scale_dtype = op->scale_dtype;
    a.out_f32 = op->out_glue == GPTQHIP_OUT_PARTIAL_F32 ? 1 : 0;
    a.in_glue = op->in_glue;
    a.glue_b = op->norm_weight;
    a.residual = op->residual;
    a.eps = op->eps;
    a.stats_in = op->stats_in;
    a.stats_n = op->stats_n;
    a.stats_out = op->stats_out;
    a.out_glue = op->out_glue == GPTQHIP_OUT_PARTIAL_F32 ? GPTQHIP_OUT_NONE : op->out_glue;
    return launch_skinny(a, pl, ws ? reinterpret_cast<float*>(ws + L.slabs_off) : nullptr,
                         ws ? reinterpret_cast<int*>(ws + L.counters_off) : nullptr, reinterpret_cast<hipStream_t>(stream));
}

int gptqhip_decode_linear_seq(const gptqhip_decode_op* const* ops, int n, gptqhip_stream_t stream) {
    if (!ops || n < 0) {
        set_error("gptqhip_decode_linear_seq: null op list");
        return GPTQHIP_EINVAL;
    }
    for (int i = 0; i < n; ++i) {
        const int rc = gptqhip_decode_linear(ops[i], stream);
        if (rc) return rc;
    }
    return GPTQHIP_OK;
}

int gptqhip_dequant(const int32_t* qweight, const int32_t* qzeros, const void* scales, const int32_t* g_idx,
                    void* out, int K, int N, int group_size, int bits, int scale_dtype, int out_dtype,
                    gptqhip_stream_t stream) {
    if (!qweight || !qzeros || !scales || !out) {
        set_error("gptqhip_dequant: null tensor pointer");
        return GPTQHIP_EINVAL;
    }
    if (bits != 4 && bits != 8) {
        set_error("gptqhip_dequant: bits=%d unsupported", bits);
        return GPTQHIP_EINVAL;
    }
    const int pf = 32 / bits;
    if (K <= 0 || N <= 0 || K % pf != 0 || N % pf != 0 || group_size <= 0 || K % group_size != 0) {
        set_error("gptqhip_dequant: bad shape K=%d N=%d group_size=%d", K, N, group_size);
        return GPTQHIP_EINVAL;
    }
    return launch_dequant(qweight, qzeros, scales, g_idx, out, K, N, group_size, bits, scale_dtype, out_dtype,
                          reinterpret_cast<hipStream_t>(stream));
}

int gptqhip_repack_awq(const int32_t* qweight_awq, const int32_t* qzeros_awq, int32_t* qweight_out,
                       int32_t* qzeros_out, int K, int N, int G, gptqhip_stream_t stream) {
    if (!qweight_awq || !qzeros_awq || !qweight_out || !qzeros_out) {
        set_error("gptqhip_repack_awq: null tensor pointer");
        return GPTQHIP_EINVAL;
    }
    if (K <= 0 || N <= 0 || G <= 0 || K % 8 != 0 || N % 8 != 0) {
        set_error("gptqhip_repack_awq: bad shape K=%d N=%d G=%d", K, N, G);
        return GPTQHIP_EINVAL;
    }
    return launch_repack_awq(qweight_awq, qzeros_awq, qweight_out, qzeros_out, K, N, G,
                             reinterpret_cast<hipStream_t>(stream));
}

int gptqhip_widen_codes(const int32_t* qweight, const int32_t* qzeros, int32_t* qweight_out, int32_t* qzeros_out, int K, int N, int G,
                        int bits, int planar, gptqhip_stream_t stream) {
    if (!qweight || !qzeros || !qweight_out || !qzeros_out) {
        set_error("gptqhip_widen_codes: null tensor pointer");
        return GPTQHIP_EINVAL;
    }
    if (bits < 2 || bits > 8 || K <= 0 || N <= 0 || G <= 0 || K % 32 != 0 || N % 32 != 0 || (planar != 0 && planar != 1) ||
        ((bits == 5 || bits == 6 || bits == 7) && !planar)) {
        set_error("gptqhip_widen_codes: bad args K=%d N=%d G=%d bits=%d planar=%d (bits 2..8, K and N multiples of 32, 5 / 6 / 7 bits "
                  "exist only planar)", K, N, G, bits, planar);
        return GPTQHIP_EINVAL;
    }
    if ((long long)K * (bits <= 4 ? 4 : 8) / 32 > 65535 || G > 65535) {      // one grid row per output word row / per group
        set_error("gptqhip_widen_codes: K=%d G=%d exceed one launch (K * wide / 32 and G at most 65535)", K, G);
        return GPTQHIP_EINVAL;
    }
    return launch_widen_codes(qweight, qzeros, qweight_out, qzeros_out, K, N, G, bits, planar, reinterpret_cast<hipStream_t>(stream));
}

int gptqhip_dequant_tiled(const uint32_t* qweight_t, const uint32_t* meta, const int32_t* perm, void* out, int K,
                          int N, int group_size, int bits, int scale_dtype, int out_dtype, gptqhip_stream_t stream) {
    if (!qweight_t || !meta || !out) {
        set_error("gptqhip_dequant_tiled: null tensor pointer");
        return GPTQHIP_EINVAL;
    }
    int rc = validate_common("gptqhip_dequant_tiled", K, N, group_size, bits);
    if (rc) return rc;
    return launch_dequant_tiled(qweight_t, meta, perm, out, K, N, group_size, bits, scale_dtype, out_dtype,
                                reinterpret_cast<hipStream_t>(stream));
}

int gptqhip_embedding(const int64_t* ids, const uint32_t* qweight_t, const uint32_t* meta, const int32_t* inv_perm,
                      void* out, int32_t* status, int T, int K, int N, int group_size, int bits, int scale_dtype,
                      gptqhip_stream_t stream) {
    if (T == 0) return GPTQHIP_OK;
    if (!ids || !qweight_t || !meta || !out || !status || T < 0 || T > 65535) {
        set_error("gptqhip_embedding: bad arguments (T=%d, at most 65535 ids per call)", T);
        return GPTQHIP_EINVAL;
    }
    int rc = validate_common("gptqhip_embedding", K, N, group_size, bits);
    if (rc) return rc;
    return launch_embedding(ids, qweight_t, meta, inv_perm, out, status, T, K, N, group_size, bits, scale_dtype,
                            reinterpret_cast<hipStream_t>(stream));
}

static int pack_args_ok(const char* who, int K, int N, int G, int bits, int planar) {
    const bool planar_only = bits == 5 || bits == 6 || bits == 7;
    if (bits < 2 || bits > 8 || K <= 0 || N <= 0 || G <= 0 || K % 32 != 0 || N % 32 != 0 || (planar != 0 && planar != 1) ||
        (planar_only && !planar)) {
        set_error("%s: bad args K=%d N=%d G=%d bits=%d planar=%d (bits 2..8, K and N multiples of 32, 5 / 6 / 7 bits exist only planar)",
                  who, K, N, G, bits, planar);
        return 0;
    }
    return 1;
}

int gptqhip_pack_gptq(const float* weight, const float* scales, const int32_t* zeros, const int32_t* g_idx,
                      int32_t* qweight, int32_t* qzeros, int K, int N, int G, int bits, int planar, gptqhip_stream_t stream) {
    if (!weight || !scales || !zeros || !g_idx || !qweight || !qzeros) {
        set_error("gptqhip_pack_gptq: null tensor pointer");
        return GPTQHIP_EINVAL;
    }
    if (!pack_args_ok("gptqhip_pack_gptq", K, N, G, bits, planar)) return GPTQHIP_EINVAL;
    if (K / 4 > 65535 || G > 65535) {       // one grid row per packed row (8-bit: K / 4) / per group
        set_error("gptqhip_pack_gptq: K=%d G=%d exceed one launch", K, G);
        return GPTQHIP_EINVAL;
    }
    return launch_pack_gptq(weight, scales, zeros, g_idx, qweight, qzeros, K, N, G, bits, planar,
                            reinterpret_cast<hipStream_t>(stream));
}

int gptqhip_pack_gptq_host(const float* weight, const float* scales, const int32_t* zeros, const int32_t* g_idx,
                           int32_t* qweight, int32_t* qzeros, int K, int N, int G, int bits, int planar, int threads) {
    if (!weight || !scales || !zeros || !g_idx || !qweight || !qzeros) {
        set_error("gptqhip_pack_gptq_host: null pointer");
        return GPTQHIP_EINVAL;
    }
    if (!pack_args_ok("gptqhip_pack_gptq_host", K, N, G, bits, planar)) return GPTQHIP_EINVAL;
    for (int k = 0; k < K; ++k) {
        const int g = g_idx[k] < 0 ? g_idx[k] + G : g_idx[k];
        if (g < 0 || g >= G) {
            set_error("gptqhip_pack_gptq_host: g_idx[%d]=%d is out of range for groups=%d", k, g_idx[k], G);
            return GPTQHIP_EINVAL;
        }
    }
    if (bits == 2 || bits == 4 || bits == 8) planar = 0;      // planar words of these widths ARE the continuous ones
    const int groups = K / 32;                                  // 32 rows -> `bits` packed rows (gptqhip_codes.h)
    const float maxq = (float)((1 << bits) - 1);
    auto work = [&](int g0, int g1) {
#pragma clang fp contract(off)      // multiply, then add, like the reference: no fma (matters on hosts compiled with FMA enabled)
        uint32_t c[32], out[8];
        for (int grp = g0; grp < g1; ++grp) {
            for (int n = 0; n < N; ++n) {
                for (int i = 0; i < 32; ++i) {
                    const int k = grp * 32 + i;
                    const int g = g_idx[k] < 0 ? g_idx[k] + G : g_idx[k];
                    float scale = scales[(size_t)g * N + n];
                    const float offset = (float)zeros[(size_t)g * N + n] * scale;
                    if (scale == 0.0f) scale = 1e-6f;
                    float q = nearbyintf((weight[(size_t)n * K + k] + offset) / scale);
                    c[i] = (uint32_t)(int)std::max(0.0f, std::min(q, maxq));
                }
                gptqhip::encode_group32(c, bits, planar, out);
                for (int t = 0; t < bits; ++t) qweight[((size_t)grp * bits + t) * N + n] = (int32_t)out[t];
            }
        }
    };
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, groups));
    std::vector<std::thread> pool;
    const int per = (groups + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int g0 = t * per, g1 = std::min(groups, g0 + per);
        if (g0 < g1) pool.emplace_back(work, g0, g1);
    }
    for (auto& th : pool) th.join();
    const uint32_t mask = (1u << bits) - 1u;
    const size_t zcols = (size_t)N * bits / 32;
    for (int g = 0; g < G; ++g)
        for (int cg = 0; cg < N / 32; ++cg) {
            uint32_t c[32], out[8];
            for (int i = 0; i < 32; ++i) c[i] = (uint32_t)zeros[(size_t)g * N + cg * 32 + i] & mask;
            gptqhip::encode_group32(c, bits, planar, out);
            for (int t = 0; t < bits; ++t) qzeros[(size_t)g * zcols + (size_t)cg * bits + t] = (int32_t)out[t];
        }
    return GPTQHIP_OK;
}

int gptqhip_gather_cols(const void* x, const int32_t* perm, void* out, int M, int K, gptqhip_stream_t stream) {
    if (M == 0) return GPTQHIP_OK;
    if (!x || !perm || !out || M < 0 || K <= 0) {
        set_error("gptqhip_gather_cols: bad args");
        return GPTQHIP_EINVAL;
    }
    return launch_gather_cols(x, perm, out, M, K, reinterpret_cast<hipStream_t>(stream));
}

int gptqhip_rmsnorm_gather(const void* h, const void* weight, const int32_t* perm, void* out, int M, int K, float eps, int act_dtype,
                           gptqhip_stream_t stream) {
    if (M == 0) return GPTQHIP_OK;
    if (!h || !weight || !out || M < 0 || K <= 0 || K % 8 != 0 || K > 16384 || h == out) {
        set_error("gptqhip_rmsnorm_gather: bad args (K a multiple of 8 up to 16384, out must not alias h)");
        return GPTQHIP_EINVAL;
    }
    if (act_dtype != GPTQHIP_FP16 && act_dtype != GPTQHIP_BF16) {
        set_error("gptqhip_rmsnorm_gather: act_dtype must be GPTQHIP_FP16/BF16");
        return GPTQHIP_EINVAL;
    }
    return launch_rmsnorm_gather(h, weight, perm, out, M, K, eps, act_dtype, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
