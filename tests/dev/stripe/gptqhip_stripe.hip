// Stripe kernel (gptqhip_stripe_kernel.h), host side: planner, launcher, CPU self-check of the partition arithmetic, and the
// instantiations for one (activation dtype, scale dtype) pair per translation unit (-DGPTQHIP_STRIPE_ACT / _SCL select which;
// gptqhip_stripe.hip without them builds the shared host code + the fp16/fp16 kernels).
#include "gptqhip_stripe_kernel.h"

#include <stdlib.h>

#include <algorithm>
#include <vector>

#ifndef GPTQHIP_STRIPE_ACT
#define GPTQHIP_STRIPE_ACT 0
#define GPTQHIP_STRIPE_SCL 0
#define GPTQHIP_STRIPE_HOST 1
#endif

namespace gptqhip {

// pipeline depth: as many stages as fit 144 KiB of LDS, at most 3 (a stage = KG x MT*16 rows x 256 B)
constexpr int stripe_depth(int mt, int kg) { return (144 * 1024) / (kg * mt * 4096) >= 3 ? 3 : 2; }

#define GPTQHIP_STRIPE_FN_(a, s) launch_stripe_a##a##_s##s
#define GPTQHIP_STRIPE_FN(a, s) GPTQHIP_STRIPE_FN_(a, s)

template <int ACT, int SCL, int GPC, int MT, int KG>
static int launch_one(const StripeParams& p, hipStream_t stream) {
    hipLaunchKernelGGL((stripe_kernel<ACT, SCL, GPC, MT, KG, stripe_depth(MT, KG)>), dim3(kStripeGrid), dim3(512), 0, stream, p);
    return check_hip(hipGetLastError(), "stripe_kernel launch");
}

template <int ACT, int SCL, int GPC>
static int launch_mt(const StripeParams& p, int mt, int kg, hipStream_t stream) {
    if (kg == 2) {
        switch (mt) {
            case 2: return launch_one<ACT, SCL, GPC, 2, 2>(p, stream);
            case 4: return launch_one<ACT, SCL, GPC, 4, 2>(p, stream);
            case 6: return launch_one<ACT, SCL, GPC, 6, 2>(p, stream);
            case 8: return launch_one<ACT, SCL, GPC, 8, 2>(p, stream);
        }
    } else {
        switch (mt) {
            case 4: return launch_one<ACT, SCL, GPC, 4, 1>(p, stream);
            case 8: return launch_one<ACT, SCL, GPC, 8, 1>(p, stream);
            case 12: return launch_one<ACT, SCL, GPC, 12, 1>(p, stream);
            case 16: return launch_one<ACT, SCL, GPC, 16, 1>(p, stream);
        }
    }
    set_error("stripe kernel: no instantiation for %d row tiles with %d K-groups", mt, kg);
    return -22;
}

int GPTQHIP_STRIPE_FN(GPTQHIP_STRIPE_ACT, GPTQHIP_STRIPE_SCL)(const StripeParams& p, int gpc, int mt, int kg, hipStream_t stream) {
    if (gpc == 1) return launch_mt<GPTQHIP_STRIPE_ACT, GPTQHIP_STRIPE_SCL, 1>(p, mt, kg, stream);
    return launch_mt<GPTQHIP_STRIPE_ACT, GPTQHIP_STRIPE_SCL, 4>(p, mt, kg, stream);
}

#ifdef GPTQHIP_STRIPE_HOST
int launch_stripe_a0_s0(const StripeParams& p, int gpc, int mt, int kg, hipStream_t stream);
int launch_stripe_a1_s0(const StripeParams& p, int gpc, int mt, int kg, hipStream_t stream);
int launch_stripe_a0_s1(const StripeParams& p, int gpc, int mt, int kg, hipStream_t stream);
int launch_stripe_a1_s1(const StripeParams& p, int gpc, int mt, int kg, hipStream_t stream);

// Shape of the launch.  KG = 2 (64-column stripes, two K-groups per block) keeps the contributors of a stripe few on narrow layers
// (N / 64 stripes over 8 queues of 32 items: 4096 columns -> 4 items per stripe) and therefore the last arriver's serial slab
// reads short; wide layers have more stripes than items anyway and take KG = 1 (128-column stripes: half the activation traffic
// per column, each of the 8 waves a full chunk per stage).  Rows beyond one panel (128 / 256) become further row panels = further
// virtual stripes of the same columns (their weights come from the L2 the second time).
StripePlan plan_stripe(int M, int K, int N, int group_size, int bits, int force_kg, int force_items) {
    StripePlan pl;
    pl.ok = 0;
    if (bits != 4 || M < 1 || K % kChunkK != 0 || group_size % 32 != 0) return pl;
    const int chunks = K / kChunkK;
    // (tuning: force_kg = kg + 10 * max row tiles per panel, e.g. 41 = 64-row panels on 128-column stripes)
    const int force_mt = force_kg / 10;
    force_kg %= 10;
    int kg = N >= 8192 ? 1 : 2;
    if (force_kg == 1 || force_kg == 2) kg = force_kg;
    if (chunks % kg != 0) kg = 1;
    int max_mt = kg == 2 ? 8 : 16;
    const int step_mt = kg == 2 ? 2 : 4;
    if (force_mt >= step_mt && force_mt <= max_mt && force_mt % step_mt == 0) max_mt = force_mt;
    const int panels = ceil_div(M, max_mt * 16);
    const int rows = ceil_div(M, panels);                    // rows per panel, balanced
    const int mt = ceil_div(ceil_div(rows, 16), step_mt) * step_mt;
    pl.kg = kg;
    pl.mt = mt;
    pl.panels = ceil_div(M, mt * 16);
    pl.gpc = (group_size % kChunkK == 0) ? 1 : 4;
    const int bn = (8 / kg) * kTileN;
    pl.stripes = ceil_div(N, bn);
    pl.vstripes = pl.stripes * pl.panels;
    pl.sps = chunks / kg;
    pl.nq = kStripeQueues;
    pl.items = kStripeItemsPerQueue;
    if (force_items >= 2 && force_items <= kStripeItemsPerQueue) pl.items = force_items;   // (tuning: fewer, longer items)
    pl.slab_floats = (size_t)pl.nq * pl.items * 2 * (size_t)(mt * 16 * bn);
    if (pl.slab_floats * 4 >= ((size_t)1 << 31)) return pl;
    // 32-bit partition arithmetic in the kernel (stripe_item_start / stripe_item_of / stripe_queue_range)
    const size_t max_u = (size_t)ceil_div(pl.vstripes, pl.nq) * pl.sps;
    if ((max_u + 1) * pl.items >= ((size_t)1 << 31) || (size_t)pl.vstripes * (pl.nq + 1) >= ((size_t)1 << 31)) return pl;
    pl.ok = 1;
    return pl;
}

int launch_stripe(const GemmArgs& a, const StripePlan& pl, float* slabs, int* heads, int* tickets, int write_through, hipStream_t stream) {
    StripeParams p;
    TiledParams& t = p.t;
    t.x = a.x;
    t.qw = a.qweight;
    t.meta = a.meta;
    t.bias = a.bias;
    t.out = a.out;
    t.M = a.M;
    t.K = a.K;
    t.N = a.N;
    t.ldo = a.ldo > 0 ? a.ldo : a.N;
    t.G = a.K / a.group_size;
    t.group_size = a.group_size;
    t.chunks = ceil_div(a.K, kChunkK);
    t.tiles = ceil_div(a.N, kTileN);
    t.out_f32 = 0;
    t.splits = 1;
    t.chunks_per_split = t.chunks;
    t.slabs = nullptr;
    t.cpg_shift = -1;
    if (a.group_size % kChunkK == 0) {
        const int cpg = a.group_size / kChunkK;
        if ((cpg & (cpg - 1)) == 0) {
            int sh = 0;
            while ((1 << sh) < cpg) ++sh;
            t.cpg_shift = sh;
        }
    }
    if ((size_t)t.tiles * t.chunks * 1024 >= ((size_t)1 << 31) || (size_t)t.tiles * t.G * 64 >= ((size_t)1 << 31)) {
        set_error("stripe kernel: packed weights of one layer must stay below 2 GiB (32-bit buffer offsets)");
        return -22;
    }
    p.vstripes = pl.vstripes;
    p.panels = pl.panels;
    p.sps = pl.sps;
    p.nq = pl.nq;
    p.items = pl.items;
    p.write_through = write_through;
    p.slabs = slabs;
    p.heads = heads;
    p.tickets = tickets;
    p.stamps = nullptr;
#ifdef GPTQHIP_STRIPE_STAMPS
    {
        const char* e = getenv("GPTQHIP_STRIPE_STAMP_PTR");   // dev builds: device buffer of [grid][16] uint64, address in hex
        if (e && *e) p.stamps = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 16));
    }
#endif
    if (a.act_dtype == kFP16 && a.scale_dtype == kFP16) return launch_stripe_a0_s0(p, pl.gpc, pl.mt, pl.kg, stream);
    if (a.act_dtype == kBF16 && a.scale_dtype == kFP16) return launch_stripe_a1_s0(p, pl.gpc, pl.mt, pl.kg, stream);
    if (a.act_dtype == kFP16 && a.scale_dtype == kBF16) return launch_stripe_a0_s1(p, pl.gpc, pl.mt, pl.kg, stream);
    return launch_stripe_a1_s1(p, pl.gpc, pl.mt, pl.kg, stream);
}

// CPU walk of exactly the arithmetic the kernel runs (same inline functions): every (virtual stripe, step) is covered once, every
// multi-contributor stripe gets distinct slab slots that the reducer's slot rule finds again, contributors are consecutive items.
// Returns 0 when consistent, else a negative code with the message in gptqhip_last_error(); *max_contrib = the largest number of
// blocks that sum into one stripe.
int stripe_selfcheck(const StripePlan& pl, int* max_contrib) {
    const StripeGeom g = {pl.vstripes, pl.sps, pl.nq, pl.items};
    std::vector<int> cover((size_t)pl.vstripes * pl.sps, 0);
    int maxc = 0;
    for (int q = 0; q < pl.nq; ++q) {
        int v0, nvs;
        stripe_queue_range(g, q, v0, nvs);
        const int U = nvs * pl.sps;
        const int P = stripe_items_of_queue(pl.items, U);
        std::vector<std::vector<int>> slots(nvs);   // slab slot ids published per stripe
        for (int item = 0; item < P; ++item) {
            const int s0 = stripe_item_start(item, U, P), s1 = stripe_item_start(item + 1, U, P);
            if (U > 0 && s0 < s1 && (stripe_item_of(s0, U, P) != item || stripe_item_of(s1 - 1, U, P) != item)) {
                set_error("stripe selfcheck: item_of disagrees with item_start (queue %d item %d)", q, item);
                return -1;
            }
            for (int s = s0; s < s1;) {
                const int t = s / pl.sps, ub = s - t * pl.sps;
                const int ue = std::min(pl.sps, ub + (s1 - s));
                const int i_first = stripe_item_of(t * pl.sps, U, P), i_last = stripe_item_of((t + 1) * pl.sps - 1, U, P);
                if (item < i_first || item > i_last) {
                    set_error("stripe selfcheck: item %d outside its stripe's contributor range [%d, %d]", item, i_first, i_last);
                    return -2;
                }
                for (int u = ub; u < ue; ++u) ++cover[(size_t)(v0 + t) * pl.sps + u];
                if (i_last > i_first) {
                    const int slot = (q * pl.items + item) * 2 + (s == s0 ? 0 : 1);
                    // what the reducer computes for this contributor
                    const int started_here = stripe_item_start(item, U, P) >= t * pl.sps;
                    if (slot != (q * pl.items + item) * 2 + (started_here ? 0 : 1)) {
                        set_error("stripe selfcheck: reducer would look in the wrong slab (queue %d item %d stripe %d)", q, item, t);
                        return -3;
                    }
                    slots[t].push_back(slot);
                }
                maxc = std::max(maxc, i_last - i_first + 1);
                s += ue - ub;
            }
        }
        for (int t = 0; t < nvs; ++t) {
            const int i_first = stripe_item_of(t * pl.sps, U, P), i_last = stripe_item_of((t + 1) * pl.sps - 1, U, P);
            const int want = i_last > i_first ? i_last - i_first + 1 : 0;
            std::vector<int> sl = slots[t];
            std::sort(sl.begin(), sl.end());
            if ((int)sl.size() != want || std::adjacent_find(sl.begin(), sl.end()) != sl.end()) {
                set_error("stripe selfcheck: stripe %d of queue %d has %zu published slabs, expected %d distinct ones", t, q, sl.size(), want);
                return -4;
            }
        }
        // slabs of different stripes must not collide either (an item publishes at most two: first and last segment)
        std::vector<int> all;
        for (auto& sv : slots) all.insert(all.end(), sv.begin(), sv.end());
        std::sort(all.begin(), all.end());
        if (std::adjacent_find(all.begin(), all.end()) != all.end()) {
            set_error("stripe selfcheck: two segments of queue %d share a slab", q);
            return -5;
        }
    }
    for (size_t i = 0; i < cover.size(); ++i)
        if (cover[i] != 1) {
            set_error("stripe selfcheck: step %zu covered %d times", i, cover[i]);
            return -6;
        }
    if (max_contrib) *max_contrib = maxc;
    return 0;
}
#endif

}  // namespace gptqhip
