/*
 * gptqhip.h -- C ABI of libgptqhip.so: the MI355X (gfx950 / CDNA4) GPTQ/AWQ grouped int4/int8
 * dequant-matmul backend.  This is the whole drop-in boundary: plain pointers and sizes, no torch
 * types.  The only callers are the two QuantLinear classes in gptqmodel_amd/nn_modules/qlinear/
 * (HipGptqLinear, HipAwqLinear) through ctypes (gptqmodel_amd/_lib.py); INTEGRATION.md shows the
 * binding a GPTQModel maintainer would add.
 *
 * Each entry point names the reference interface it replaces (paths relative to the upstream
 * ModelCloud/GPTQModel tree).  Conventions:
 *   - every call returns 0 on success, a negative GPTQHIP_E* code on failure; the message is
 *     available from gptqhip_last_error() (thread-local).  The Python wrapper raises RuntimeError,
 *     mirroring TORCH_CHECK -> RuntimeError of the reference natives
 *     (gptqmodel_ext/exllamav2/ext_gptq.cpp:31-60).
 *   - all tensor pointers are DEVICE pointers on the current HIP device; the module owns them, the
 *     library borrows them for the duration of the call (no opaque handles; contrast the leaked
 *     QMatrix* handle of ext_gptq.cpp:73-93).
 *   - work is enqueued on `stream` (the caller passes torch.cuda.current_stream().cuda_stream, as
 *     the reference natives use the current stream: ext_gptq.cpp:108); nothing synchronises.
 *   - re-entrant across threads, devices and streams provided each (device, stream) uses its own workspace
 *     (error string and tuning overrides are thread-local; there is no other mutable state).
 *
 * Checkpoint ("canonical", GPTQ v2 K-packed) layout accepted by gptqhip_repack_tiled / gptqhip_dequant:
 *   qweight int32 [K*bits/32, N]   word (r,n) holds codes k = pf*r + j at bits [bits*j, bits*j+bits)
 *   qzeros  int32 [G, N*bits/32]   word (g,c) holds zero  n = pf*c + j at bits [bits*j, ...)   (v2: used as-is)
 *   scales  fp16|bf16 [G, N]
 * = the buffer contract of gptqmodel/nn_modules/qlinear/__init__.py:827-865 after the loader's
 *   v1->v2 conversion (gptqmodel/utils/model.py:750-844).  AWQ checkpoints are first brought to this
 *   layout by gptqhip_repack_awq.
 *
 * Kernel ("tiled", MFMA-tile-major) layout consumed by gptqhip_gemm, produced ONCE in post_init by
 * gptqhip_repack_tiled (the reference's fast kernels repack in post_init too: marlin.py:246-293,
 * exllamav2.py:114-140):
 *   qweight_t uint32 [ceil(N/16)][ceil(K/128)][64 lanes][4]   (x2 for 8-bit)  -- one (tile, chunk) block is
 *             the B operand of four mfma_f32_16x16x32 steps and exactly one 1 KiB wave load; see
 *             gptqmodel_amd/csrc/gptqhip_device.h for the nibble order.
 *   meta      uint32 [ceil(N/16)][G][16] = scale16 | (0xE400|zero)<<16  (pre-baked dequant constants)
 *   act-order: rows are stored group-sorted (row k' = checkpoint row perm[k'], perm = stable argsort(g_idx));
 *             gptqhip_gemm gathers x through the same perm.
 */
#ifndef GPTQHIP_H
#define GPTQHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPTQHIP_ABI_VERSION 19

/* error codes */
#define GPTQHIP_OK 0
#define GPTQHIP_EINVAL (-22)   /* bad argument / unsupported shape       */
#define GPTQHIP_ENOMEM (-12)   /* workspace too small                    */
#define GPTQHIP_EHIP (-5)      /* a HIP runtime call failed              */
#define GPTQHIP_ENODEV (-19)   /* no gfx950 device                       */

/* dtype tags for activations / scales */
#define GPTQHIP_FP16 0
#define GPTQHIP_BF16 1

/* opaque stream: a hipStream_t passed as void* so that callers need no HIP headers */
typedef void* gptqhip_stream_t;

int gptqhip_abi_version(void);
const char* gptqhip_last_error(void);

/* Device probe used by QuantLinear.validate_once() (gptqmodel/nn_modules/qlinear/__init__.py:257-270):
 * fills CU count, bytes of HBM and the gcn arch name; returns GPTQHIP_ENODEV unless the device is gfx950. */
int gptqhip_device_info(int device, int* cu_count, size_t* hbm_bytes, char* arch, int arch_len);

/* Bytes of zero-initialised device scratch gptqhip_gemm needs for this problem (split-K fp32 slabs +
 * arrival counters + act-order gather buffer).  Precedent: ExllamaV2 per-device ScratchSpace,
 * gptqmodel/utils/model.py:1304-1313.  The workspace must be zero-filled once at allocation; the
 * kernels leave it zeroed where it matters (counters).  Takes the same (group_size, bits) gptqhip_gemm will be
 * called with: both sides run the same launch planner, so the size can never disagree with the launch.  The size is NOT monotonic
 * in M (the planner may split K at one batch size and not at a larger one): size a shared workspace by the maximum over the M values
 * that will be used, not by the largest M.  0 = bad args. */
size_t gptqhip_workspace_bytes(int M, int K, int N, int group_size, int bits, int has_perm);

/* Sizes (in 32-bit words) of the tiled weight / meta arrays for a [K,N] layer. */
size_t gptqhip_tiled_words(int K, int N, int bits);
size_t gptqhip_meta_words(int K, int N, int group_size);

/* post_init: checkpoint layout -> tiled layout + meta.  perm [K] int32 or NULL (act-order: sorted row k' is
 * checkpoint row perm[k']).  Same role as gptq_marlin_repack (gptqmodel_ext/marlin/gptq_marlin_repack.cu:250)
 * and ExllamaV2 make_sequential/shuffle (gptqmodel_ext/exllamav2/cuda/q_matrix.cu:19-45,502-604).
 *   bits 4|8; K % 32 == 0; N % 8 == 0; group_size % 32 == 0 and K % group_size == 0.
 * qweight and qweight_t may both be NULL: only the meta constants are (re)built (e.g. scales cast to a new dtype). */
int gptqhip_repack_tiled(const int32_t* qweight, const int32_t* qzeros, const void* scales, const int32_t* perm,
                         uint32_t* qweight_t, uint32_t* meta, int K, int N, int group_size, int bits,
                         gptqhip_stream_t stream);

/* THE HOT PATH.  out[M,N] = x[M,K] @ dequant(qweight_t, meta) (+ bias), rounded like the reference:
 *   W = round_scaledtype(scale * (code - zero)); W' = round_actdtype(W); y = round_actdtype(sum_k x*W');
 *   y = round_actdtype(y + bias).
 * Replaces TorchLinear.forward/_forward_eager (gptqmodel/nn_modules/qlinear/torch.py:302-347 with
 * _dequantize_weight_cached_248 :700-717) and AwqTorchLinear.forward (torch_awq.py:157-195 with
 * dequantize_gemm, quantization/awq/utils/packing_utils.py:106-121).
 *   x        [M,K]  act_dtype, row-major contiguous
 *   perm     [K] int32 or NULL: x column gather for act-order
 *   bias     [N] act_dtype or NULL
 *   out      [M,N] act_dtype
 *   scale_dtype: dtype of the scale bits inside meta
 *   flags    GPTQHIP_GEMM_PARTIAL_F32: `out` is float32 [M,N] and receives the UNROUNDED fp32 accumulators (bias
 *            must be NULL) -- the partial sums a row-parallel (K-sharded) tensor-parallel layer all-reduces before
 *            rounding once, so TP reproduces the single-GPU rounding chain.
 *            GPTQHIP_GEMM_EXACT_BF16 (opt-in, default off): bf16 activations, 4-bit weights, group_size % 128 == 0,
 *            M <= 4: accumulate the exact products s*(q-z)*x instead of first rounding every weight to bf16 like
 *            torch.py:326-335 does (gfx950 has no packed bf16 VALU; 7 instead of 16 VALU per packed word).  The result is
 *            the exact-arithmetic value, up to 2 output ulps away from the reference's chain; ignored elsewhere
 *            (fp16 activations included: measured there, it does not pay -- docs/history/DESIGN_rounds_1-5.md 4.1.1). */
#define GPTQHIP_GEMM_PARTIAL_F32 1
#define GPTQHIP_GEMM_EXACT_BF16 2
int gptqhip_gemm(const void* x, const uint32_t* qweight_t, const uint32_t* meta,
                 const int32_t* perm, const void* bias, void* out,
                 void* workspace, size_t workspace_bytes,
                 int M, int K, int N, int group_size, int bits,
                 int act_dtype, int scale_dtype, int flags, gptqhip_stream_t stream);

/* BATCH-1 DECODE OP with fused decoder-layer glue (the skinny kernel's M = 1 pipeline, gptqmodel_amd/csrc/gptqhip_skinny.hip).
 *
 * One call = one quantised linear of a decode step, out[N] = glue_out( glue_in(x)[K] @ dequant(qweight_t, meta) + bias ):
 * the same contraction and rounding chain as gptqhip_gemm at M = 1 (TorchLinear.forward, torch.py:302-347), plus the
 * elementwise ops a Llama-style decoder layer runs between its linears, so that no separate glue launches (each one a
 * dependent-kernel boundary of ~2-3 us at batch 1) sit on the token's critical path:
 *   in_glue  GPTQHIP_GLUE_NONE      x [K]
 *            GPTQHIP_GLUE_RMSNORM   x [K] = the residual stream h; the kernel feeds norm_weight * act(h32 * rsqrt(mean(h32^2)
 *                                   + eps)) (HF LlamaRMSNorm: fp32 statistics, rounded to the activation dtype, then * weight)
 *            GPTQHIP_GLUE_SILU_MUL  x [2K] = gate | up; the kernel feeds act(silu(gate)) * up (HF LlamaMLP)
 *   residual [N] or NULL: out = act(residual + y)   (hidden = residual + hidden, one more rounding)
 *   out_glue GPTQHIP_OUT_SILU_MUL_PAIRED: the layer is a fused gate|up projection whose columns were interleaved in blocks of
 *            8 (8 gate columns, the 8 matching up columns, ...: utils.model.fuse_gate_up_interleaved), so every 16-column
 *            tile holds both halves of 8 MLP neurons and the epilogue writes out[N/2] = act(silu(gate)) * up directly -- the
 *            activation is computed ONCE per element by the producer instead of by every consumer block
 *            GPTQHIP_OUT_PARTIAL_F32: out is float32 [N], the unrounded accumulators of a K-shard (tensor parallel row-parallel layer)
 *   stats_out [ceil(N/16)] floats or NULL: per-tile sum of out^2 -- the RMSNorm statistic of the op consuming `out`
 *   stats_in / stats_n: with GPTQHIP_GLUE_RMSNORM, the producer's stats_out for this op's x (stats_n = ceil(K/16) <= 512):
 *            each wave sums the partials in a fixed order (one load per lane, no block barrier); NULL: every block reduces the
 *            row itself (first op of a step, whose input comes from outside the chain)
 * The glue is applied to each ring stage's activation pair on its way into the MFMA A fragment (RMSNorm statistics are
 * reduced once per block while the first weight loads are in flight).  Stream-ordered like every other entry point.
 * Supported: shapes on the decode kernel's regular pipeline (K % 128 == 0, group_size = 128 * 2^n or 32 | 64, a wave count
 * dividing K / 128 evenly -- every Llama-3 8B / 70B shape, also tensor-parallel shards), bits 4 | 8; an act-order permutation
 * (`perm`, applied to the glued row inside the kernel) for M = 1 and group_size % 128 == 0 only;
 * gptqhip_decode_supported() tells, anything else returns GPTQHIP_EINVAL -- use gptqhip_gemm (+ separate glue) there.
 * workspace: as gptqhip_gemm (gptqhip_workspace_bytes(1, K, N, ...)); only narrow layers (cross-block split-K) touch it.
 * The glue replaces no reference interface: it mirrors what the reference's caller (HF LlamaDecoderLayer) does between
 * QuantLinear.forward calls. */
#define GPTQHIP_GLUE_NONE 0
#define GPTQHIP_GLUE_RMSNORM 1
#define GPTQHIP_GLUE_SILU_MUL 2
#define GPTQHIP_OUT_NONE 0
#define GPTQHIP_OUT_SILU_MUL_PAIRED 1
#define GPTQHIP_OUT_PARTIAL_F32 2   /* out is float32 [N]: the UNROUNDED accumulators (row-parallel tensor-parallel shard; no bias /
                                       residual / stats_out -- gptqhip_allreduce_oneshot applies them after the reduction) */
typedef struct gptqhip_decode_op {
    const uint32_t* qweight_t;   /* tiled words (gptqhip_repack_tiled)                         */
    const uint32_t* meta;        /* [tiles][G][16] constants                                   */
    const void* bias;            /* [N] act dtype or NULL                                      */
    const void* x;               /* [K] or [2K] act dtype, see in_glue                         */
    const void* norm_weight;     /* [K] act dtype (GPTQHIP_GLUE_RMSNORM) else NULL             */
    const void* residual;        /* [N] act dtype or NULL                                      */
    void* out;                   /* [N] act dtype ([N/2] with OUT_SILU_MUL_PAIRED)             */
    void* workspace;             /* zero-initialised scratch as for gptqhip_gemm, or NULL      */
    size_t workspace_bytes;
    const float* stats_in;       /* see above                                                  */
    float* stats_out;
    const int32_t* perm;         /* act-order (desc_act) permutation [K] as for gptqhip_gemm, applied to the GLUED input inside the
                                    kernel, or NULL.  Needs K * 2 bytes <= 44 KiB (else gather first).                */
    float eps;
    int K, N, group_size, bits, act_dtype, scale_dtype, in_glue, out_glue, stats_n;
    int flags;                   /* 0 or GPTQHIP_GEMM_EXACT_BF16 (the opt-in exact-arithmetic dequant, bf16 activations)                */
    int M;                       /* rows (1..16): x [M,K], residual / out [M,N], stats_in [M][stats_n], stats_out [M][ceil(N/16)].
                                    M > 1 (a few sequences, or speculative tokens of one): in_glue NONE | RMSNORM, perm NULL.   */
} gptqhip_decode_op;
int gptqhip_decode_linear(const gptqhip_decode_op* op, gptqhip_stream_t stream);
/* The same for n ops in order on one stream (one host call per dependent run of ops: o_proj -> gate_up -> down_proj of a
 * decoder layer); stops at the first error. */
int gptqhip_decode_linear_seq(const gptqhip_decode_op* const* ops, int n, gptqhip_stream_t stream);
/* 1 if gptqhip_decode_linear supports a [K,N] layer with this group size (has_perm: with an act-order
 * permutation; M: rows, 1..16), else 0. */
int gptqhip_decode_supported(int K, int N, int group_size, int has_perm, int M);

/* Materialise W[K,N] from the CHECKPOINT layout in `out_dtype` (= scales dtype in the reference).  Replaces
 * TorchLinear.dequantize_weight (torch.py:225) / PackableQuantLinear.dequantize_weight
 * (qlinear/__init__.py:947-1003); bit-exact with it.  g_idx [K] int32 or NULL (then k/group_size). */
int gptqhip_dequant(const int32_t* qweight, const int32_t* qzeros, const void* scales, const int32_t* g_idx,
                    void* out, int K, int N, int group_size, int bits, int scale_dtype, int out_dtype,
                    gptqhip_stream_t stream);

/* Same from the TILED layout (what a module holds after post_init); with perm the rows are written back in
 * checkpoint order. */
int gptqhip_dequant_tiled(const uint32_t* qweight_t, const uint32_t* meta, const int32_t* perm, void* out,
                          int K, int N, int group_size, int bits, int scale_dtype, int out_dtype,
                          gptqhip_stream_t stream);

/* post_init helper for the OTHER bit widths of the reference's generic dequantize_weight (gptqmodel/nn_modules/qlinear/__init__.py:947-999:
 * continuous 2- and 3-bit words -- the 3-bit codes 10 and 21 of every 32 straddle a word, :982-991 -- and the planar 3 / 5 / 6 / 7-bit
 * layout of gptqmodel/utils/planar_packing.py:7-24): qweight int32 [K*bits/32, N] / qzeros int32 [G, N*bits/32] -> the same codes and
 * zero-points in the continuous 4-bit (bits <= 4) or 8-bit layout every other entry point reads (qweight_out [K*wide/32, N], qzeros_out
 * [G, N*wide/32], wide = bits <= 4 ? 4 : 8).  Values are unchanged, so W = scale * (code - zero) is too; the wider copy costs HBM bytes
 * (3 bits stored as 4: +33 %), not correctness.  K % 32 == 0, N % 32 == 0; planar = 1 for 5 / 6 / 7 bits (they exist only planar) and for
 * FORMAT.GPTQ_P checkpoints of the other widths. */
int gptqhip_widen_codes(const int32_t* qweight, const int32_t* qzeros, int32_t* qweight_out, int32_t* qzeros_out,
                        int K, int N, int G, int bits, int planar, gptqhip_stream_t stream);

/* post_init helper: AWQ GEMM layout (qweight [K,N/8], qzeros [G,N/8], nibble i <-> column 8c+[0,2,4,6,1,3,5,7][i])
 * -> checkpoint-canonical layout.  Semantics of unpack_reorder_pack (packing_utils.py:90-103); zero-points kept as-is. */
int gptqhip_repack_awq(const int32_t* qweight_awq, const int32_t* qzeros_awq,
                       int32_t* qweight_out, int32_t* qzeros_out, int K, int N, int G,
                       gptqhip_stream_t stream);

/* Quantised embedding lookup: out[t, :] = dequantised row ids[t] of the [K = num_embeddings, N = dim]
 * matrix held in the TILED layout, in the scales dtype -- replaces TorchQuantEmbeddings.forward
 * (gptqmodel/nn_modules/qlinear/torch.py:764-797), which dequantises the WHOLE table on every call and then runs
 * F.embedding.  ids int64 [T]; inv_perm [K] int32 or NULL (act-order: tiled row of checkpoint row k); out [T,N] 16-bit.
 * Ids outside [0, K) are an error reported through `status` (device int32, set to 1) like torch's index check. */
int gptqhip_embedding(const int64_t* ids, const uint32_t* qweight_t, const uint32_t* meta, const int32_t* inv_perm,
                      void* out, int32_t* status, int T, int K, int N, int group_size, int bits, int scale_dtype,
                      gptqhip_stream_t stream);

/* Quantise-and-pack on the device: weight fp32 [N,K] (nn.Linear layout), scales fp32 [G,N], zeros
 * int32 [G,N], g_idx int32 [K] (negative entries wrap by +G) -> checkpoint-layout qweight int32 [K*bits/32, N] and
 * qzeros int32 [G, N*bits/32].  q = clamp(rint((w + zero*scale)/scale), 0, maxq) in fp32, scale == 0 -> 1e-6:
 * bit-exact with the reference packer pack_block_cpu (gptqmodel_ext/pack_block_cpu.cpp:105-190) and
 * PackableQuantLinear.pack_block (gptqmodel/nn_modules/qlinear/__init__.py:1036-1323).  K % 32 == 0, N % 32 == 0.
 * bits 2..8: 2 / 4 / 8 bits tile a word; 3 bits are a continuous 96-bit stream per 32 codes (planar = 0) or split-plane words
 * (planar = 1: FORMAT.GPTQ_P); 5 / 6 / 7 bits exist only planar (planar must be 1) -- the layouts of gptqhip_widen_codes' input,
 * pinned by tests/golden/ref_pack_bits.npz (the reference's pack_block at every width). */
int gptqhip_pack_gptq(const float* weight, const float* scales, const int32_t* zeros, const int32_t* g_idx,
                      int32_t* qweight, int32_t* qzeros, int K, int N, int G, int bits, int planar, gptqhip_stream_t stream);

/* The same quantise-and-pack on the HOST (all pointers are CPU memory): the C++ equivalent of the reference's native
 * packer gptqmodel::pack_block_cpu (gptqmodel_ext/pack_block_cpu.cpp:17, at::parallel_for over blocks :100),
 * threaded over groups of 32 rows with `threads` std::threads (<= 0: hardware concurrency).  Needs no GPU. */
int gptqhip_pack_gptq_host(const float* weight, const float* scales, const int32_t* zeros, const int32_t* g_idx,
                           int32_t* qweight, int32_t* qzeros, int K, int N, int G, int bits, int planar, int threads);

/* out[m, k'] = x[m, perm[k']]  (16-bit elements).  Used by gptqhip_gemm internally and exported for tests
 * (ExllamaV2 gathers A through q_perm: gptqmodel_ext/exllamav2/cuda/q_gemm_kernel_gptq.cuh:79-90). */
int gptqhip_gather_cols(const void* x, const int32_t* perm, void* out, int M, int K, gptqhip_stream_t stream);

/* RMSNorm fused with the act-order gather, for the PREFILL of act-order (desc_act) checkpoints:
 *     out[m, k'] = weight[p] * act(h32[m, p] * rsqrt(mean_k(h32[m, k]^2) + eps)),   p = perm[k']   (perm NULL: p = k')
 * HF LlamaRMSNorm arithmetic (fp32 statistics, the normalised value rounded to the activation dtype, the product with the weight
 * rounded once).  The callers of the q|k|v and gate|up QuantLinears are RMSNorms; their siblings share one g_idx, so the
 * normalised x can leave the norm kernel already in the kernel's row order and gptqhip_gemm runs with perm = NULL -- the separate
 * x gather per linear (ExllamaV2 does it while staging A: gptqmodel_ext/exllamav2/cuda/q_gemm_kernel_gptq.cuh:79-90) disappears.
 * h, out [M,K] act dtype (out must not alias h), weight [K] act dtype, K % 8 == 0, K <= 16384. */
int gptqhip_rmsnorm_gather(const void* h, const void* weight, const int32_t* perm, void* out, int M, int K, float eps, int act_dtype,
                           gptqhip_stream_t stream);

/* ONE-SHOT ALL-REDUCE for the tensor-parallel decode step (gptqmodel_amd/csrc/gptqhip_comm.hip).  The reference has no
 * tensor parallelism and no collectives -- nothing upstream is replaced; this is the MI355X design for the 70B
 * config's 160 latency-bound all-reduces per token (32 KB each at batch 1): every rank pushes its fp32 partial vector straight
 * into every rank's peer-mapped buffer over xGMI (point-to-point, no intermediate hop), waits for the `world` arrival flags in
 * its OWN buffer, sums the slots in rank order (bit-identical on all ranks) and applies the reference's rounding chain
 * (act(sum); act(+bias); act(residual + .)) in the same kernel.  Epochs live in device memory: capture-safe.
 *   gptqhip_comm_bytes(world, n_max)   bytes of one rank's communication buffer for vectors of up to n_max floats (0 = bad args;
 *                                      world <= 8, n_max <= 65536)
 *   gptqhip_comm_alloc(bytes, &ptr, handle[64])   uncached (fine-grained) device memory, zero-filled, + its IPC handle
 *   gptqhip_comm_open(handle, &ptr) / gptqhip_comm_close(ptr)   map / unmap a PEER's buffer in this process
 *   gptqhip_comm_free(ptr)             free the buffer this rank allocated
 *   gptqhip_comm_status(own_buf, &st)  host read of the sticky status word (0 = healthy; bit 0: a bounded wait for a peer gave up,
 *                                      bit 1: gptqhip_allgather_select met an index outside [0, n_local * world) -- in both cases
 *                                      the affected outputs were poisoned with NaN)
 *   gptqhip_comm_set_timeout(own_buf, ms)  bound of this rank's peer waits (host write, between launches; default 10 s or
 *                                      GPTQHIP_COMM_TIMEOUT_MS at alloc time) -- the self-test runs with a short one
 *   gptqhip_allreduce_oneshot(partial[n] fp32, peer_bufs[world] (HOST array of device pointers, own buffer at [rank]), rank, world,
 *                             n (% 4 == 0), n_max (as allocated), bias|NULL, residual|NULL, out[n] act dtype,
 *                             stats_out[ceil(n/16)]|NULL (per-16 sums of out^2: the next decode op's RMSNorm statistic), act_dtype, stream)
 *   gptqhip_allgather_select(x_local[n_local] act dtype, peer_bufs, rank, world, n_local (% 8 == 0, the same on every rank), n_max,
 *                            index[n_out] int32 | NULL, n_out, out[n_out], act_dtype, stream)
 *                                      one-shot ALL-GATHER of the ranks' 16-bit vectors + select: out[j] = concat_r(x_r)[index[j]]
 *                                      (index NULL: the whole vector, n_out = n_local * world).  The input exchange of an act-order
 *                                      row-parallel shard cut from globally group-sorted rows (the rule of
 *                                      gptqmodel/utils/marlin.py:296-305,368-372): its rows need input features scattered over all
 *                                      ranks' column shards.  Own epoch / flags / slots inside the same buffer.
 * Ordering: data stores (system scope, write-through) -> system-scope release fence per wave -> block barrier -> flag store-release;
 * the waiter polls its own buffer, then a system-scope acquire fence.  A wait that exceeds its bound (GPTQHIP_COMM_TIMEOUT_MS at
 * gptqhip_comm_alloc time, default 10 s) writes NaN to the block's outputs and sets the sticky status word: a lost peer is loud.
 * Every rank must issue the same sequence of calls.  Status: exercised by two processes sharing one GPU through real IPC
 * mappings incl. a 10^5-epoch stress (tests/test_gpu_comm.py); OneShotAllReduce.self_test() validates a communicator against the
 * process group's own collective on whatever hardware it runs on. */
#define GPTQHIP_IPC_HANDLE_BYTES 64
size_t gptqhip_comm_bytes(int world, int n_max);
int gptqhip_comm_alloc(size_t bytes, void** dev_ptr, unsigned char* handle_out);
int gptqhip_comm_open(const unsigned char* handle, void** dev_ptr);
int gptqhip_comm_close(void* dev_ptr);
int gptqhip_comm_free(void* dev_ptr);
int gptqhip_comm_status(void* own_buf, uint32_t* status_out);
int gptqhip_comm_set_timeout(void* own_buf, unsigned int timeout_ms);
int gptqhip_allreduce_oneshot(const float* partial, void* const* peer_bufs, int rank, int world, int n, int n_max,
                              const void* bias, const void* residual, void* out, float* stats_out, int act_dtype,
                              gptqhip_stream_t stream);
int gptqhip_allgather_select(const void* x_local, void* const* peer_bufs, int rank, int world, int n_local, int n_max,
                             const int32_t* index, int n_out, void* out, int act_dtype, gptqhip_stream_t stream);

/* Which kernel family and launch geometry gptqhip_gemm would use for this call, as text (triage / logging / tests; host logic, no
 * GPU): "skinny launches=1 mt=2 nt=4 waves=8 depth=2 regular=1 splits=1 gather=0", "tiled bm=64 splits=8 tail_cols=0 gather=1" or
 * "tiled bm=80 bn=128 splits=1 tail_cols=0 gather=0" (bm = rows per block tile: 32..128 in steps of 16, or 256; bn = columns per block,
 * printed when it is 128 = one 16-column tile per wave instead of two).
 * mt = 16-row tiles per block, nt = column tiles per block (4 / 2: the wide-layer form), gather = a separate act-order x gather pass
 * runs first.  The reference steers its kernels with thresholds too (ExllamaV2 switches to dequant + cuBLAS above 50 rows:
 * gptqmodel_ext/exllamav2/cuda/q_gemm.cu:118, config.h:4); here the crossover is measured per layer shape (docs/history/DESIGN_rounds_1-5.md 4.1.1). */
int gptqhip_plan_describe(int M, int K, int N, int group_size, int bits, int has_perm, char* buf, int buf_len);

/* Tuning hook (benchmarks / tests): force the cross-block split-K factor and the waves per block of the skinny
 * kernel (0 = heuristic; with the prefill kernel 1 / 2 / 3 = 256- / 128- / 64-row tiles, 32..128 in steps of 16 = that tile height, 1000 +
 * rows (1032..1128) = that height with 128-column blocks),
 * or the kernel family (0 auto, 1 skinny, 2 tiled-prefill).  The overrides are THREAD-LOCAL
 * (they apply to gptqhip_gemm / gptqhip_workspace_bytes calls made by the calling thread only), so the library keeps
 * no process-global mutable state and stays re-entrant across threads, devices and streams. */
int gptqhip_set_tuning(int force_split_k, int force_kernel, int force_waves);

/* Batch-1 decode form (round 6): which kernel serves gptqhip_gemm at M = 1 and gptqhip_decode_linear at M = 1 (4-bit, one group constant per
 * 128-row chunk, no in-kernel act-order permutation; everything else keeps skinny_kernel).  All forms compute the reference's
 * y = x @ (s * (q - z)) (TorchLinear._forward_eager, torch.py:326-347); they differ in the kernel structure and in whether every weight is
 * rounded to the scales' dtype BEFORE the contraction like the reference does (torch.py:716-717):
 *   5  DEFAULT for fp16 activations with fp16 scales and for bf16 activations (any scale dtype).  skinny1_kernel ("preload": a wave parks the glued x pieces and the group constants of
 *      all its chunks in LDS up front, so the ring carries ONE VMEM instruction per 1 KiB chunk) + RAW CODES: the 4-bit codes enter the MFMA
 *      as they are -- (w & 0x000F000F) is a pair of fp16 denormals q * 2^-24, (w & 0x00F000F0) a pair q * 2^-20; the matrix pipe takes fp16
 *      denormals at face value (tests/dev/mfma_denorm_probe.hip) -- one AND per two weights, no zero-point subtraction:
 *      y = sum_g s_g * (2^24 lo_g + 2^20 hi_g - z_g * Sx_g),  lo_g / hi_g = the fp32 sums of the two code classes of group g, Sx_g = the sum
 *      of the group's (glued) activations, taken once per wave when x is parked.  Like form 3 this is the exact-arithmetic value of the
 *      reference's expression up to fp32 accumulation; it differs from the reference's own output only by the reference's per-weight
 *      rounding fp16(s (q - z)) (2^-12 relative, random): inside north_star's 1e-3 bar on every golden and no further from float64
 *      arithmetic than the reference chain (tests/test_gpu_decode_forms.py), NOT bit-identical to forms 0 / 4.  +5 % over form 3.
 *      bf16 activations: the f16 matrix pipe is the one that takes the codes as denormals (bf16 denormals times x underflow fp32), so a wave
 *      converts its (glued) bf16 x pieces to fp16 -- exactly: 8 significant bits -- after dividing them by a power of two taken from its own
 *      largest |x| (no fp16 overflow whatever the input), and multiplies the power of two back in through the chunk constants; the output is
 *      the bf16 rounding of the exact sum.  Against the reference (every weight rounded to bf16 first) single outputs sit one bf16 ulp per
 *      rounding step away: inside the reference's own element-wise gates (tests/test_torch_kernel_accuracy.py:111-125: allclose against
 *      EXACT fp32 arithmetic, atol 3e-2 / rtol 1e-2; tests/kernels/test_gptq.py:229-266, 353-360: isclose rtol 0.15 / atol 0.008).
 *      +16 % over form 4 on the bf16 chain (878 -> 1015 tokens/s), 0.96 of fp16.
 *   3  skinny1_kernel + GROUP-FACTORED dequant (the fp16 default until form 5): the exact integers (q - z) go into the MFMA as fp16 pairs
 *      (magic-number route, 9 VALU per packed word) and the group's scale multiplies the fp32 partial sum once per chunk,
 *      y = sum_g s_g * (sum_{k in g} x_k (q_k - z_g)).  +6 % over form 4.
 *   4  DEFAULT for fp16 activations with bf16 scales, and for everything when GPTQHIP_DECODE_BITFAITHFUL=1 is in the environment: skinny1_kernel
 *      with the reference's per-weight rounding kept (same bits as form 0 up to the fp32 summation order).  +6 % over form 0.
 *   0  skinny_kernel, per-weight rounding (the rounds 1-5 kernel).
 *   2  skinny_kernel + group-factored dequant (fp16 x fp16).
 *   1  decode_stream_kernel (gptqmodel_amd/csrc/gptqhip_stream.hip): weights streamed HBM -> LDS by LDS-DMA into per-wave rings, raw code
 *      pairs (1024 + q | 64 + q) contracted on the matrix pipe, offsets / zero-points / scale taken out per chunk; fp16 x fp16 only (other
 *      dtypes fall to form 4).  Built for VERDICT r5 item 1b, measured 15-20 % SLOWER than form 0 (profiles/r06_stream_kernel_ablation.txt):
 *      opt-in, kept as the record of that experiment.
 *  -1  back to the process default.
 * THREAD-LOCAL like gptqhip_set_tuning.  The reference steers its own kernels with the same kind of switch (env flags, torch.py:172-190). */
int gptqhip_set_decode_form(int form);

#ifdef __cplusplus
}
#endif
#endif /* GPTQHIP_H */
