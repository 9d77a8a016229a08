/*
 * ngp_hip.h — C ABI of libngp_hip.so: the MI355X (gfx950) Instant-NGP hot path for JNeRF.
 *
 * One entry point per `jt.code(...)` site of the reference's hot path plus the
 * fused variants the MI355X design adds.  Conventions (mirroring the reference's operator interface,
 * SURVEY.md §8b "Ownership / Errors / Threading"):
 *   - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch-ROCm allocator here,
 *     Jittor's allocator in the reference) unless the name ends in `_host`;
 *   - nothing is allocated or freed inside; scratch is passed in by the caller;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it, no call synchronises;
 *   - return value: 0 = launched, <0 = argument error (NGP_E_*), >0 = hipError_t from the launch;
 *     ngp_last_error() gives a message for the calling thread;
 *   - `dtype` is the table / network-output element type T of the reference (`grad_t`): NGP_F32 or NGP_F16;
 *   - capacity overflow keeps the reference's silent-saturation semantics (ray_sampler.h:74-80,
 *     compacted_coord.h:63) — it is never an error.
 * File:line citations are relative to /root/reference/python/jnerf/.
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NGP_ABI_VERSION 3
enum { NGP_F32 = 0, NGP_F16 = 1 };
enum { NGP_E_ARG = -1, NGP_E_DTYPE = -2, NGP_E_ALIGN = -3, NGP_E_CAPACITY = -4 };
/* feature-tensor layouts between the encoder and the MLP */
enum { NGP_LAYOUT_AOS = 0 /* [n,32] as HashEncoder returns it */, NGP_LAYOUT_SOA = 1 /* [16][n] pairs, level-major */,
       NGP_WEIGHTS_PACKED = 0x100 /* OR into feat_layout of the field-network calls: `wd` is the fragment buffer ngp_field_pack_weights wrote, `wc` is ignored */ };

int ngp_abi_version(void);
const char *ngp_last_error(void);
/* device query used by bench.py: fills {CU count, max clock kHz, L2 bytes, total global MiB}; returns 0 */
int ngp_device_info(int device, int64_t *out4_host);
/* self-test of the MFMA fragment-layout assumptions the fused MLP relies on (A=I with asymmetric B); result u32[4] device, [0]==0 ok */
int ngp_selftest_mfma(void *stream, uint32_t *result);

/* ---- hash grid ------------------------------------------------------------------------------------------------
 * level_table_host: u32[16][4] = {offset (entries), size (entries), resolution, scale (f32 bits)} per level, built by the host
 * exactly as position_encoders/hash_encoder/grid_encode.py:17-40 + op_header/HashEncode.h:149-151 prescribe. */
/* host helper: fills level_table_host for an aabb_scale, returns m_n_params (grid_encode.py:36) */
uint32_t ngp_level_table(double aabb_scale, uint32_t *level_table_host);
/* replaces GridEncode.execute (grid_encode.py:71-125: extract_position + kernel_grid + transpose_encoded_position) */
int ngp_hash_encode_fwd(void *stream, uint32_t n, const float *pos, uint32_t pos_stride_floats, const void *table,
                        const uint32_t *level_table_host, void *out, int dtype, int out_layout, const uint32_t *n_valid /*device u32 or NULL*/);
/* The same forward with d(encoding)/d(position): the `dy_dx` output of the reference's kernel_grid (op_header/HashEncode.h:205-251), which grid_encode.py:96 leaves
 * disabled (`float*dy_dx=nullptr`).  dy_dx: f32[n][3][32], dy_dx[i][d][2*level+f] = d out[i][2*level+f] / d pos[i][d] (the reference's layout, :249).  This is what a
 * hash-grid SDF network (BASELINE configs[4]) needs from the encoder. */
int ngp_hash_encode_fwd_dydx(void *stream, uint32_t n, const float *pos, uint32_t pos_stride_floats, const void *table, const uint32_t *level_table_host,
                             void *out, int dtype, int out_layout, const uint32_t *n_valid, float *dy_dx);
/* dL/dpos[i][d] = sum_k dLdy[i][k] * dy_dx[i][d][k] -> f32[n,3].  GridEncode.grad returns None for the positions (grid_encode.py:190) and the reference has no kernel
 * for this contraction: it is what that `None` would have to become. */
int ngp_hash_encode_bwd_input(void *stream, uint32_t n, const void *dLdy, int dtype, int in_layout, const float *dy_dx, float *dLdpos, const uint32_t *n_valid);
/* Second-order terms for a network trained on its own input gradient (NeuS over a hash-grid SDF network, BASELINE configs[4]: the eikonal term and the normal fed to
 * the colour network, python/jnerf/models/samplers/neus_render/renderer.py:214,258-260 through neus_network.py:99-108, back-propagate through dL/dpos).  For an
 * upstream gradient u = d loss / d (dL/dpos), f32[n,3]:
 *   ..._bwd_dy:   ddLdy[i][k] = sum_d u[i][d] * dy_dx[i][d][k]   (AoS [n,32], dtype f32|f16) - the gradient w.r.t. the dLdy that ngp_hash_encode_bwd_input was given;
 *   ..._bwd_grid: grad[entry][f] += dLdy[i][2 level + f] * sum_d u[i][d] * d w_corner / d pos_d   (fp32 atomics into the table gradient, NOT zeroed first) - the
 *                 gradient w.r.t. the table that dy_dx was computed from.  dLdy is AoS [n,32].
 * The reference has no counterpart (its dy_dx branch is compiled but never enabled, grid_encode.py:96); the gradient w.r.t. pos itself is not produced. */
int ngp_hash_encode_bwd_input_bwd_dy(void *stream, uint32_t n, const float *u, const float *dy_dx, void *ddLdy, int dtype);
int ngp_hash_encode_bwd_input_bwd_grid(void *stream, uint32_t n, const float *pos, uint32_t pos_stride_floats, const void *dLdy, int dtype, const float *u,
                                       const uint32_t *level_table_host, float *grad, uint64_t n_params);
/* replaces GridEncode.grad (grid_encode.py:137-184: memset + transpose_gradients + kernel_grid_backward).
 * grad_dtype may be NGP_F32 with dtype NGP_F16 (fp32 accumulation of fp16 gradients). zero_first!=0 clears `grad` (n_params elements). */
int ngp_hash_encode_bwd(void *stream, uint32_t n, const float *pos, uint32_t pos_stride_floats, const void *dLdy, const uint32_t *level_table_host,
                        void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid);

/* Same contract with a caller-provided workspace - the training path.  No float atomic anywhere, every level through records that one workgroup per bin of table
 * entries sums in exact 64-bit integer arithmetic in LDS: order-independent, bit-reproducible, the gradient is OVERWRITTEN entry by entry (no memset).  Routed by the level
 * table and the dtypes (csrc/hash_encode.hip: hash_bwd_path):
 *   dtype == grad_dtype == NGP_F32 (ngp_base.py): record REGIONS - run-combined 12-byte records for the levels up to resolution 300, one 16-byte edge record per cell
 *     edge for the finer ones, one accumulate kernel (r4);
 *   fp16 dL/dy (ngp_fox.py): per-corner record lists (6-byte fp16 records for the fine levels) with cursor reservations (r2 / r3).
 * A level table the bins cannot take (a level beyond 2^19 entries, a hashed table that is not a power of two) runs the reference's scheme - one global float atomic per
 * corner - as ngp_hash_encode_bwd without a workspace does: the ONE fallback of this stage (r5: the owner-computes scan of rounds 1-2 and its fixed-point entry point
 * ngp_hash_encode_bwd_fx are gone).
 * workspace: >= ngp_hash_bwd_workspace_bytes_for(level table, n, dtype, grad_dtype) bytes (fp32: ~0.6 GB at n = 2^18, fp16: ~1.7 GB); a smaller one is NGP_E_CAPACITY
 * (rounds 2-4 silently took a slower path).  ngp_hash_bwd_workspace_bytes(level table, n) = the larger of the two, for a caller that serves both. */
uint64_t ngp_hash_bwd_workspace_bytes(const uint32_t *level_table_host, uint32_t n);
uint64_t ngp_hash_bwd_workspace_bytes_for(const uint32_t *level_table_host, uint32_t n, int dtype, int grad_dtype);
int ngp_hash_encode_bwd_ws(void *stream, uint32_t n, const float *pos, uint32_t pos_stride_floats, const void *dLdy, const uint32_t *level_table_host,
                           void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid,
                           void *workspace, uint64_t workspace_bytes);

/* ---- direction encoding: replaces SHEncoder.execute (position_encoders/sh_encoder/sh_encoder.py:29-51, SphericalEncode.h:45-95) */
int ngp_sh_encode(void *stream, uint32_t n, const float *dir, uint32_t dir_stride_floats, void *out /*[n,16]*/, int dtype);

/* ---- field network -------------------------------------------------------------------------------------------------
 * Both MLPs of NGPNetworks.execute_ (models/networks/ngp_network.py:77-84) in one fp16-MFMA kernel, weights resident in LDS:
 * replaces mlp_fused_forward_func x2 (ops/code_ops/fully_fused_mlp.py:52-86) + SHEncoder + the three concats.
 * wd: f16[3072] = W0[64x32] W1[16x64];  wc: f16[7168] = V0[64x32] V1[64x64] V2[16x64]  — the FMLP pack (ngp_network.py:21-29).
 * feat: f16 features in `feat_layout`; dir: f32 warped directions; out: [n,4] T = (r,g,b logits, log-density).
 * The kernels read the weights as MFMA-ordered fragments.  By default every call builds them from wd/wc (a 5 us kernel, into a per-stream scratch);
 * a caller that runs forward and backward on the same weights builds them once with ngp_field_pack_weights (f16[NGP_PACKED_WEIGHT_HALVES], 16-byte
 * aligned) and passes feat_layout | NGP_WEIGHTS_PACKED. */
#define NGP_PACKED_WEIGHT_HALVES 21504
int ngp_field_pack_weights(void *stream, const void *wd, const void *wc, void *packed_out);
int ngp_field_fwd(void *stream, uint32_t n, const void *feat, int feat_layout, const float *dir, uint32_t dir_stride_floats,
                  const void *wd, const void *wc, void *out, int out_dtype, const uint32_t *n_valid);
/* NGPNetworks.density (ngp_network.py:86-89): density MLP only, out [n] T (column 0) */
int ngp_density_fwd(void *stream, uint32_t n, const void *feat, int feat_layout, const void *wd, void *out, int out_dtype);
/* replaces mlp_fused_backward_func x2 + the 5 cublas_acc_matmul wgrads (fully_fused_mlp.py:93-145): forward is recomputed in-kernel.
 * dLdout [n,4] T -> dLdfeat f16 (feat_layout) ; weight-gradient partial slabs f32[n_slabs][10240] (wd part first), reduced by ngp_reduce_slabs. */
int ngp_field_bwd(void *stream, uint32_t n, const void *feat, int feat_layout, const float *dir, uint32_t dir_stride_floats,
                  const void *wd, const void *wc, const void *dLdout, int out_dtype, void *dLdfeat, float *wgrad_slabs, uint32_t n_slabs,
                  const uint32_t *n_valid);
int ngp_field_bwd_slabs(uint32_t n);                       /* number of slabs ngp_field_bwd writes for capacity n */
int ngp_reduce_slabs(void *stream, const float *slabs, uint32_t n_slabs, uint32_t width, float *out /*[width]*/, int accumulate /*out += sum*/);

/* ---- fp32 field network: NGPNetworks.execute_ when cfg.fp16 is unset (ngp_network.py:57-67 falls back to nn.Linear chains in fp32; this is what
 * projects/ngp/configs/ngp_base.py - the lego headline - runs).  Same fused structure on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate).
 * wd f32[3072] / wc f32[7168]: the same pack layout as above; feat / dLdfeat f32 in `feat_layout`; out / dLdout f32 [n,4]; slabs as ngp_field_bwd.
 * ngp_field32_pack_weights builds the MFMA-ordered fragments once per step (f32[NGP_PACKED32_WEIGHT_FLOATS], 16-byte aligned; pass with NGP_WEIGHTS_PACKED). */
#define NGP_PACKED32_WEIGHT_FLOATS 40960   /* 76 fp32 fragments of 256 floats (forward + transposed) + 2 x 42 split fp16 fragments of 512 halves (csrc/field_split.h) */
int ngp_field32_pack_weights(void *stream, const float *wd, const float *wc, float *packed_out);
int ngp_field32_fwd(void *stream, uint32_t n, const float *feat, int feat_layout, const float *dir, uint32_t dir_stride_floats,
                    const float *wd, const float *wc, float *out, const uint32_t *n_valid);
int ngp_density32_fwd(void *stream, uint32_t n, const float *feat, int feat_layout, const float *wd, float *out /*[n]*/);
int ngp_field32_bwd(void *stream, uint32_t n, const float *feat, int feat_layout, const float *dir, uint32_t dir_stride_floats,
                    const float *wd, const float *wc, const float *dLdout, float *dLdfeat, float *wgrad_slabs, uint32_t n_slabs, const uint32_t *n_valid);
int ngp_field32_bwd_slabs(uint32_t n);
/* The default kernels behind the three calls above compute every fp32 product from split fp16 operands (csrc/field_split.hip); operands are lifted into fp16's range by
 * fixed powers of two, so features beyond ~255 or activations beyond ~4094 would overflow - the reference's fp32 nn.Linear chain (ngp_network.py:59-67) has no such limit.
 * The kernels therefore track the largest operand they split and raise a device-side flag: bit 0 = an operand came within 4x of the limit (nothing has overflowed yet),
 * bit 1 = one left the range (that launch's results contain infinities) or a backward stored a non-finite feature gradient.
 * ngp_field32_range_check returns the flag bits (>= 0; reset != 0 clears them) or NGP_E_ARG; it reads 4 bytes back - call it where the host synchronises anyway.
 * ngp_field32_select(1) switches this process to the exact-product kernels (v_mfma_f32_16x16x4_f32: no operand range), (0) back to the split ones; returns the previous choice. */
int ngp_field32_range_check(int reset);
int ngp_field32_select(int exact);

/* ---- sampler ------------------------------------------------------------------------------------------------------
 * rng_state_host: u64[2] = {state, inc} of the reference's global pcg32{1337} (ops/code_ops/global_vars.py:13-16); it is advanced
 * by 2^32 on return exactly like `rng.advance()` at ray_sampler.py:61.  cascades = NERF_CASCADES (5), const_dt per cfg.const_dt. */
/* replaces RaySampler.execute (samplers/density_grid_sampler/ray_sampler.py:34-62, op_header/ray_sampler.h:4-114). Deterministic: slots are
 * reserved in ray order (what the reference's atomicAdd gives under a serial launch). counters: u32[2] = {rays with a slot, steps reserved}.
 * scratch: u32[n_rays + 1024]. coords [max_samples,7] is zero-filled first when zero_coords!=0 (ray_sampler.py:50). */
int ngp_march_rays(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                   float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                   float *coords, uint32_t *numsteps /*[n,2]*/, uint32_t *counters, int32_t *ray_indices, uint32_t *scratch, int zero_coords);
/* replaces CompactedCoord.execute (compacted_coord.py:39-64, op_header/compacted_coord.h:4-76); counter u32[1]; scratch u32[n_rays+1024] */
int ngp_compact_coords(void *stream, uint32_t n_rays, uint32_t cap, const float *coords_in, const uint32_t *numsteps_in, float *coords_out,
                       uint32_t *numsteps_out, uint32_t *counter, uint32_t *scratch);
/* MI355X training path: march + compaction in one pass (the reference's dead forward pass is dropped, SURVEY.md App.B-1).
 * Result == ngp_compact_coords(ngp_march_rays(...)) for the same inputs; rows >= min(total,cap) of coords_out are NOT touched
 * (consumers take counters[2] as n_valid). counters: u32[4] = {rays, steps reserved, compacted steps (unclamped), min(steps,cap)}.
 * scratch: u32[ngp_march_scratch_elems(n_rays)]. */
uint64_t ngp_march_scratch_elems(uint32_t n_rays);   /* u32 elements of `scratch` ngp_march_rays_compacted needs (step counts + per-ray t-cache) */
int ngp_march_rays_compacted(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                             float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                             uint32_t cap, float *coords_out, uint32_t *numsteps, uint32_t *numsteps_compacted, uint32_t *counters, uint32_t *scratch);
/* same, and also writes the warped positions as a compact f32[cap,3] array (what ngp_hash_encode_fwd/_bwd stream 16 times per step); pos_out may be NULL */
int ngp_march_rays_compacted_pos(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                                 float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                                 uint32_t cap, float *coords_out, uint32_t *numsteps, uint32_t *numsteps_compacted, uint32_t *counters, uint32_t *scratch, float *pos_out);

/* same, with the occupied bounds of the bitfield (the buffer ngp_grid_occupied_bounds fills; NULL = off): rays that cannot meet an occupied cell are dropped before
 * the traversal and every ray stops behind the last occupied box.  Results are IDENTICAL to the call without bounds - a sample is only ever emitted inside an occupied cell -
 * the traversal just no longer evaluates the ~1400 candidates of a ray that sees nothing but background (most rays of an object-centred scene's training batch). */
int ngp_march_rays_compacted_bounds(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                                    float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                                    uint32_t cap, float *coords_out, uint32_t *numsteps, uint32_t *numsteps_compacted, uint32_t *counters, uint32_t *scratch, float *pos_out,
                                    const int32_t *occ_bounds);

/* replaces CalcRgb.execute / .grad / .inference (calc_rgb.py:45-68, 78-104, 120-144; op_header/calc_rgb.h) */
int ngp_composite_fwd(void *stream, uint32_t n_rays, const void *net_out, int dtype, const float *coords, const uint32_t *numsteps,
                      const uint32_t *numsteps_compacted, const float *bg /*[n,3]*/, int cascades, float *rgb_out);
/* ngp_composite_fwd followed by ngp_huber on the composited colours (runner.py:72 with HuberLoss), one launch: loss (may be NULL) and loss_grad are [n_rays,3] */
int ngp_composite_fwd_huber(void *stream, uint32_t n_rays, const void *net_out, int dtype, const float *coords, const uint32_t *numsteps,
                            const uint32_t *numsteps_compacted, const float *bg, int cascades, float *rgb_out, const float *target, float delta, float *loss, float *loss_grad);
int ngp_composite_bwd(void *stream, uint32_t n_rays, uint32_t n_elems, const void *net_out, int dtype, const float *coords,
                      const uint32_t *numsteps_compacted, const float *loss_grad, const float *rgb_ray, const float *density_grid_mean,
                      int cascades, void *dLdout, int zero_first);
/* (r5) ngp_composite_fwd_huber followed by ngp_composite_bwd (zero_first = 0) as ONE launch - what the native training step issues; same arguments, bit-identical rgb,
 * loss, loss_grad and dLdout.  n_elems: rows of net / dLdout (the sample capacity of the batch: selects the lanes-per-ray variant like ngp_composite_bwd does).  (r6) For
 * arguments on which the two split launches would pick DIFFERENT variants (n_elems != 2^18 with a ray count between their thresholds) the call issues those two launches. */
int ngp_composite_train(void *stream, uint32_t n_rays, uint32_t n_elems, const void *net, int dtype, const float *coords, const uint32_t *numsteps, const uint32_t *numsteps_compacted,
                        const float *bg, int cascades, float *rgb_out, const float *target, float huber_delta, float *loss, float *loss_grad, const float *density_grid_mean,
                        void *dLdout);
int ngp_composite_inference(void *stream, uint32_t n_rays, const void *net_out, int dtype, const float *coords, const uint32_t *numsteps,
                            int cascades, float *rgb_out, float *alpha_out);
/* HuberLoss (models/losses/huber_loss.py:6-14) value [n] and its elementwise derivative (= autograd of the summed loss) */
int ngp_huber(void *stream, uint32_t n, const float *x, const float *target, float delta, float *loss, float *grad);

/* ---- density grid (every update_den_freq steps; density_grid_sampler.py:204-264) --------------------------------------------- */
int ngp_grid_mark_untrained(void *stream, uint32_t n_elements, float *grid, uint32_t n_images, const float *focal /*[n,2]*/,
                            const float *xforms /*[n,4,3]*/, int W, int H);
int ngp_grid_generate_samples(void *stream, uint32_t n, uint64_t *rng_state_host, const uint32_t *ema_step /*device*/, float aabb0, float aabb1,
                              const float *grid, float *positions /*[n,3]*/, uint32_t *indices, uint32_t n_cascades, float thresh);
/* same samples (the same multiset of (position, cell) pairs); morton_order != 0 stores them so that consecutive slots hold consecutive Morton cells: the
 * 16-level gather of the density query that follows then walks the tables coherently (the consumer, ngp_grid_splat_max, is order-independent) */
int ngp_grid_generate_samples_ordered(void *stream, uint32_t n, uint64_t *rng_state_host, const uint32_t *ema_step /*device*/, float aabb0, float aabb1,
                                      const float *grid, float *positions /*[n,3]*/, uint32_t *indices, uint32_t n_cascades, float thresh, int morton_order);
int ngp_grid_splat_max(void *stream, uint32_t n, const uint32_t *indices, const void *density /*[n] T*/, int dtype, float *grid_tmp);
int ngp_grid_ema(void *stream, uint32_t n_elements, float decay, float *grid, const float *grid_tmp);
/* mean over cascade 0 -> mean[0]; grid_to_bitfield; 4x bitfield_max_pool (update_bitfield.py:15-37) */
int ngp_grid_update_bitfield(void *stream, const float *grid, int cascades, float *mean /*[1]*/, uint8_t *bitfield);
/* (ours) where the occupied cells are, for the marcher's culling (ngp_march_rays_compacted_bounds / NgpRenderChunk.occ_bounds).  `bounds` is a device buffer of
 * NGP_OCC_BOUNDS_INTS i32: [6 c .. 6 c + 5] = {min x, y, z, max x, y, z} of cascade c's occupied cells (min > max: empty cascade); from int NGP_OCC_COARSE_OFFSET_INTS on,
 * NGP_OCC_COARSE^3 bytes: the dilated 32^3 coarse map of the unit cube (cascades 0 and 1), then the same number of scratch bytes.  Call after ngp_grid_update_bitfield, on the
 * same stream. */
#define NGP_OCC_COARSE 32
#define NGP_OCC_COARSE_OFFSET_INTS 64
#define NGP_OCC_BOUNDS_INTS (NGP_OCC_COARSE_OFFSET_INTS + 2 * NGP_OCC_COARSE * NGP_OCC_COARSE * NGP_OCC_COARSE / 4)
int ngp_grid_occupied_bounds(void *stream, const uint8_t *bitfield, int cascades, int32_t *bounds);

/* ---- optimiser: Adam (optims/adam.py + Jittor nn.Adam) -> ExpDecay lr (host) -> EMA.ema_step (optims/ema.py:26-37), one sweep.
 * p/m/v/ema are fp32 masters; p_half (may be NULL) receives the fp16 copy the kernels gather from; g is fp32 or fp16 (g_dtype) and is
 * zeroed for the next step when zero_grad!=0.  step is 1-based.  ema may be NULL (no EMA), a separate buffer, or == p: the caller declares that the
 * stored EMA equals the parameter (true after every ema_step, ema.py:37 `v <- p`), which saves 8 B/parameter of traffic. */
/* data parallel: fp32 gradient -> fp16 buffer for the RCCL all-reduce (half the bytes over xGMI; the reference keeps fp16 gradients anyway), source optionally
 * zeroed in the same pass.  n % 8 == 0, 16-byte aligned.  Feed the reduced fp16 buffer to ngp_adam_ema_step with g_dtype = NGP_F16. */
int ngp_grad_to_half(void *stream, uint64_t n, float *grad_f32, void *grad_f16, int zero_src);
/* same with the values multiplied by `scale` (a power of two) before the conversion: gradients carry the 128 / n_rays loss scale and would otherwise sit in
 * fp16's subnormal range; the sweep undoes it (ngp_adam_ema_step_scaled, grad_mul = 1 / scale) */
int ngp_grad_to_half_scaled(void *stream, uint64_t n, float *grad_f32, void *grad_f16, int zero_src, float scale);
int ngp_adam_ema_step(void *stream, uint64_t n, float *p, void *g, int g_dtype, float *m, float *v, float *ema, void *p_half,
                      float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay, int zero_grad);

int ngp_adam_ema_step_scaled(void *stream, uint64_t n, float *p, void *g, int g_dtype, float *m, float *v, float *ema, void *p_half,
                             float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay, int zero_grad, float grad_mul /* fp16 gradients are multiplied by this */);

/* ---- ray generation (dataset/dataset.py:172-188) + target compositing (runner/runner.py:66-68) ------------------------------- */
int ngp_generate_rays(void *stream, uint32_t n, const int64_t *pixel_index, int W, int H, const float *focal, const float *metadata /*[n_img,11]*/,
                      const float *xforms, const float *images /*[n_img*H*W,4] or NULL*/, const float *bg /*[n,3] or NULL*/,
                      int32_t *img_id, float *rays_o, float *rays_d, float *target /*[n,3] or NULL*/);

/* ---- data parallelism: ray batches shard over the GPUs of one node, gradients are summed over RCCL / xGMI (north_star).  The reference has no
 * collective call sites (its only multi-process hook is utils/general.py:39-40), so these have no jt.code counterpart; a Jittor host would call them from the same
 * place this repo's optimiser does (optims/adam.py: between backward and the parameter update).  RCCL is bound with dlopen at the first call (librccl.so.1 - the
 * copy the host framework already mapped - or $NGP_RCCL_PATH); the library has no link-time dependency on it.  Return codes: 1000 + ncclResult_t for RCCL errors. */
#define NGP_COMM_ID_BYTES 128
#define NGP_DP_COARSE_RES_MAX 300           /* levels up to this resolution form the first bucket (== the scatter's run-combined levels) */
/* rank 0: ncclGetUniqueId into id_out_host[NGP_COMM_ID_BYTES]; the host distributes it (this repo: a broadcast over the host framework's process group) */
int ngp_comm_unique_id(void *id_out_host);
/* every rank, with its device current: ncclCommInitRank.  *comm_out is an opaque handle for the calls below */
int ngp_comm_init(void **comm_out, int rank, int world, const void *id_host);
int ngp_comm_destroy(void *comm);
/* ncclCommAbort: also terminates the communicator's collectives still in flight on the device (a communicator whose self-test never completed). */
int ngp_comm_abort(void *comm);
int ngp_comm_rank_world(void *comm, int *rank_out, int *world_out);
/* SUM all-reduce, in place, of n_bufs gradient buffers (device pointers in a HOST array; counts in elements; dtypes NGP_F32 | NGP_F16) as one RCCL group on `stream` */
int ngp_allreduce_grads(void *comm, void *stream, int n_bufs, void *const *bufs_host, const uint64_t *counts_host, const int *dtypes_host);
/* How a table of n_params elements is dealt to `world` ranks: up to two buckets [cut[b], cut[b+1]) whose boundaries are multiples of 8*world elements, each
 * split into `world` equal shards (rank r owns [cut[b] + r*shard_count[b], +shard_count[b])), and a replicated tail [tail_begin, n_params) of < 8*world elements.
 * With n_buckets == 2 the boundary is the first element of the first level finer than NGP_DP_COARSE_RES_MAX, rounded down (cut_level = that level).  Pure host code. */
typedef struct NgpDpPlan {
	uint64_t cut[3]; uint64_t shard_begin[2], shard_count[2]; uint64_t tail_begin, tail_count;
	uint32_t n_buckets; int32_t cut_level; int32_t world, rank;
} NgpDpPlan;
int ngp_dp_plan(const uint32_t *level_table_host, uint64_t n_params, int world, int rank, int n_buckets, NgpDpPlan *out_host);
/* all-gather, in place, of every rank's shards of n_bufs buffers laid out like the table (parameters every step; masters and Adam moments before a checkpoint) */
int ngp_dp_allgather(void *comm, void *stream, const NgpDpPlan *plan_host, int n_bufs, void *const *bufs_host, const int *dtypes_host);

/* ---- NeuS: the SDF -> opacity -> weights -> colour chain of NeuSRenderer.render_core (python/jnerf/models/samplers/neus_render/renderer.py:216-252; ~25 Jittor tensor ops
 * and their autograd there), one wavefront per ray.  Rays have a fixed number of sections: n inside the unit sphere (sdf, cosv = ray . SDF gradient, dists, color[.,3],
 * inside = 1|0: f32[n_rays, n]) and, with a background model (bg_alpha / bg_color: f32[n_rays, n_total(,3)], NULL without), n_total - n more behind them; inside the
 * first n the background takes over where inside == 0 (:238-244).  inv_s: DEVICE scalar (the variance network's exp(10 v), clipped).  n_total <= 512.
 * fwd writes colour f32[n_rays,3], weights / alpha f32[n_rays,n_total] and p, c f32[n_rays,n] (the 'p', 'c' / 'cdf' entries of render_core's result).
 * bwd takes dL/dcolour and (or NULL) dL/dweights and writes dL/d{sdf, cosv, color} like their inputs, dL/d{bg_alpha, bg_color} (when given) and one partial of
 * dL/dinv_s per ray (the caller sums them).  The [0,1] clip of the opacity is Jittor's safe_clip: value clamped, gradient passed through. */
int ngp_neus_composite_fwd(void *stream, uint32_t n_rays, uint32_t n, uint32_t n_total, const float *sdf, const float *cosv, const float *dists, const float *inv_s,
                           const float *color, const float *inside, const float *bg_alpha, const float *bg_color, float cos_anneal_ratio,
                           float *out_color, float *weights, float *alpha, float *p, float *c);
int ngp_neus_composite_bwd(void *stream, uint32_t n_rays, uint32_t n, uint32_t n_total, const float *sdf, const float *cosv, const float *dists, const float *inv_s,
                           const float *color, const float *inside, const float *bg_alpha, const float *bg_color, float cos_anneal_ratio,
                           const float *g_color, const float *g_weights, float *d_sdf, float *d_cos, float *d_inv_s_partial, float *d_color, float *d_bg_alpha, float *d_bg_color);

/* ---- one training iteration's launch sequence, issued from native code -------------------------------------------------------------------
 * The body of Runner.train for one already-sampled batch (runner/runner.py:71-76: model(pos, dir) -> sampler.rays2rgb -> HuberLoss ->
 * optimizer.step -> ema_optimizer.ema_step), i.e. exactly these calls in this order on `stream`:
 *   ngp_field_pack_weights, ngp_hash_encode_fwd, ngp_field_fwd, ngp_composite_fwd_huber, ngp_composite_bwd, ngp_field_bwd, ngp_reduce_slabs,
 *   ngp_hash_encode_bwd_ws, then (run_optimizer != 0) ngp_adam_ema_step for each of the n_opt parameter tensors.
 * It exists because eleven separate FFI crossings per iteration cost the host more than the GPU needs for the work; it adds no arithmetic
 * of its own.  Level-major features.  Data-parallel callers hand over a communicator (the exchange step then runs inside, see `comm` below) or split the call in phases. */
struct NgpDpPlan;
typedef struct NgpTrainStep {
	uint32_t n;                 /* sample capacity of the batch buffers (2^18) */
	uint32_t n_rays;
	int32_t cascades;
	int32_t run_optimizer;
	/* the sampled batch (ngp_march_rays_compacted_pos, ngp_generate_rays) */
	const float *coords;        /* [n,7] */
	const float *pos;           /* [n,3] */
	const uint32_t *numsteps, *numsteps_compacted;   /* [n_rays,2] */
	const uint32_t *n_valid;    /* device count of valid samples */
	const float *bg, *target;   /* [n_rays,3] */
	const float *density_grid_mean;
	/* hash grid: `table` is what the gather reads - the fp16 shadow (dtype NGP_F16) or the fp32 parameter itself (NGP_F32) */
	const void *table; const uint32_t *level_table_host; float *table_grad; uint64_t n_params; void *hash_workspace; uint64_t hash_workspace_bytes;
	/* field network: weight packs, fragment scratch (f16[NGP_PACKED_WEIGHT_HALVES] | f32[NGP_PACKED32_WEIGHT_FLOATS]), features / their gradient T[16][n][2], outputs T[n,4] */
	const void *wd, *wc; void *packed_weights; void *feat, *dfeat; void *out, *dout;
	float *wgrad_slabs; uint32_t n_slabs; int32_t dtype /* NGP_F16: fp16 table shadow + fused fp16-MFMA network (ngp_fox.py); NGP_F32: fp32 table + fp32-MFMA network (ngp_base.py) */; float *wgrad_flat /* f32[10240] */;
	/* loss */
	float huber_delta; float pad1; float *rgb, *loss, *loss_grad;   /* [n_rays,3] */
	/* optimiser: Adam + EMA over n_opt (<= 4) parameter tensors, see ngp_adam_ema_step */
	int32_t n_opt; uint32_t step;
	float lr, beta0, beta1, eps, ema_decay, pad2;
	float *p[4], *g[4], *m[4], *v[4], *ema[4]; void *p_half[4]; uint64_t numel[4];
	/* measurement: -1 = none, else the stage to bracket with a HIP event pair on `stream` (NGP_STAGE_*); durations are read with ngp_train_step_timings */
	int32_t timed_stage;
	/* != 0: the gradient buffers are OVERWRITTEN by this step's backward (hash scatter with zero_first, slab reduction without accumulation) and the sweep does
	 * not zero them afterwards - 4 B/parameter less traffic than accumulate-then-zero.  0: gradients are accumulated into and zeroed by the sweep. */
	int32_t grad_overwrite;
	/* (r6, no layout change) With grad_overwrite != 0, phase NGP_PHASE_ALL, run_optimizer != 0 and no communicator / plan (one GPU), the hash table's sweep is applied by
	 * the hash backward's accumulate kernel(s) in place of the gradient store (same update on the same values - bit-identical parameters and moments): the contents of
	 * table_grad after such a call are UNSPECIFIED (the reference's optimizer.step(loss) exposes no gradient either).  A caller that wants the table's gradient runs
	 * NGP_PHASE_BACKWARD (complete in table_grad) and NGP_PHASE_SWEEP, or sets NGP_NO_ADAM_RIDE=1 in the environment (A/B hook). */
	/* ---- (ABI 2) phases and data parallelism.  phase: NGP_PHASE_ALL = the whole iteration; NGP_PHASE_BACKWARD = everything up to and including the hash
	 * scatter (gradients complete in table_grad / wgrad_flat, no sweep); NGP_PHASE_SWEEP = only the Adam+EMA sweeps.  A host that owns its own collective (gloo
	 * in the two-ranks-on-one-GPU tests, Jittor's MPI hooks) calls BACKWARD, reduces the two gradient buffers, calls SWEEP.
	 * comm != NULL (ngp_comm_init; phase must be NGP_PHASE_ALL): the exchange step runs inside the call on `stream` through RCCL - reduce-scatter of the table
	 * gradient per `dp` (ngp_dp_plan), all-reduce of its tail and of wgrad_flat, sweep of this rank's shard (+ tail + every other tensor, replicated), all-gather
	 * of the updated shard of p[dp_table] (dp_gather_master != 0) and of p_half[dp_table] (when present).  With dp_gather_master == 0 (fp16 mode: the kernels read
	 * the shadow) the fp32 master, m and v are valid on their owner's shard only until ngp_dp_allgather collects them (checkpoint time).
	 * grad_wire != NULL: the table gradient travels as fp16 multiplied by wire_scale (a power of two; ngp_grad_to_half_scaled into grad_wire, f16[n_params]);
	 * the sweep divides it out.  dp_overlap != 0: two buckets - the coarse levels' reduce-scatter is issued on the library's communication stream as soon as
	 * their accumulate launch has finished, under the fine levels' accumulate (costs four event packets per iteration; off by default). */
	int32_t phase, dp_overlap, dp_table /* index into p[]/g[] of the hash table */, dp_gather_master;
	void *comm; const struct NgpDpPlan *dp; void *grad_wire; float wire_scale;
	/* != 0: packed_weights already holds the fragments of the CURRENT weights - the previous ngp_train_step's sweep wrote them (fp32 network with the flat 10240-float pack
	 * among the optimiser tensors: its Adam+EMA sweep and the fragment packing are one launch) and nothing has changed the weights since - so this call skips its packing launch */
	int32_t frags_fresh;
	/* wait_flag != NULL: the call starts with ngp_flag_wait(stream, wait_flag, wait_value, wait_status) - the batch was produced on another stream that ended it with
	 * ngp_flag_signal (cheaper than an event hand-over between two HIP streams, see below) */
	const uint32_t *wait_flag; uint32_t *wait_status; uint32_t wait_value; uint32_t pad4;
} NgpTrainStep;
enum { NGP_PHASE_ALL = 0, NGP_PHASE_BACKWARD = 1, NGP_PHASE_SWEEP = 2 };
enum { NGP_STAGE_PACK = 0, NGP_STAGE_HASH_FWD, NGP_STAGE_FIELD_FWD, NGP_STAGE_COMPOSITE_FWD, NGP_STAGE_COMPOSITE_BWD, NGP_STAGE_FIELD_BWD, NGP_STAGE_REDUCE_SLABS,
       NGP_STAGE_HASH_BWD, NGP_STAGE_ADAM /* the largest parameter tensor's sweep */,
       NGP_STAGE_BOUNDARY /* not a stage: from the end of one call's last launch to the start of the next call's first launch (main-stream idle + waits) */ };
int ngp_train_step(void *stream, const NgpTrainStep *args_host);

/* ---- hand-over between streams by device flag: ngp_flag_signal is a one-thread launch that stores `value` to *flag (device u32) - issued on the producing stream
 * BEHIND the producing kernels, so they have completed and released their writes; ngp_flag_wait is a one-wavefront launch that returns once (int32)(*flag - value) >= 0,
 * ahead of the consuming kernels on their stream (which acquire at their own launch).  The wait is bounded (2 s): on expiry it ORs 1 into *status (device u32, may be
 * NULL) and lets the stream continue rather than hang the GPU - the host must check status.  No reference counterpart (the reference runs on one stream and calls
 * .sync() after every op, SURVEY.md §8b "Threading/streams"). */
int ngp_flag_signal(void *stream, uint32_t *flag, uint32_t value);
int ngp_flag_wait(void *stream, const uint32_t *flag, uint32_t value, uint32_t *status);

/* ---- one inference chunk from native code --------------------------------------------------------------------------------------------------------------
 * The loop body of Runner.render_img (runner/runner.py:211-226: sampler.sample -> model -> rays2rgb(inference)) for `n_rays` rays: march + compaction
 * (ngp_march_rays_compacted_pos with cap = sample capacity of the buffers), hash encode, fused field network (weights pre-packed by the caller), inference
 * compositing straight into the image buffers.  No host read-back: `totals` (device u64[2]) accumulates {samples rendered, chunks whose requested samples
 * exceeded max_samples - the caller re-renders those images in smaller chunks}.  The Python loop around it spent as long on tensor plumbing as the kernels ran. */
typedef struct NgpRenderChunk {
	uint32_t n_rays, cap /* samples the buffers below hold */, max_samples;
	int32_t const_dt, cascades, dtype /* NGP_F16 | NGP_F32: table + fused network precision */;
	float aabb0, aabb1, near_distance, cone_angle;
	const float *rays_o, *rays_d;            /* [n_rays,3] */
	const uint8_t *bitfield;
	uint64_t *rng_state_host;                /* pcg32 {state, inc}; advanced like ray_sampler.py:61 */
	float *coords;                           /* [cap,7] */
	float *pos;                              /* [cap,3] */
	uint32_t *numsteps, *numsteps_compacted; /* [n_rays,2] */
	uint32_t *counters;                      /* u32[4] */
	uint32_t *scratch;                       /* ngp_march_scratch_elems(n_rays) */
	const void *table; const uint32_t *level_table_host;
	const void *packed_weights;              /* ngp_field_pack_weights / ngp_field32_pack_weights output */
	void *feat;                              /* T[16][cap][2] */
	void *out;                               /* T[cap,4] */
	float *rgb_out, *alpha_out;              /* [n_rays,3], [n_rays,1]: this chunk's rows of the image */
	uint64_t *totals;                        /* device u64[2], accumulated */
	const int32_t *occ_bounds;               /* (ABI 2) ngp_grid_occupied_bounds of `bitfield`, or NULL */
} NgpRenderChunk;
int ngp_render_chunk(void *stream, const NgpRenderChunk *args_host);
/* waits for the bracketed launches of earlier ngp_train_step calls (this thread's device) and writes up to `max` durations in milliseconds, oldest first;
 * returns how many were written (<0 on error) and forgets them */
int ngp_train_step_timings(float *ms_out_host, int max);

/* ---- measurement: per-kernel HIP-event brackets (bench.py's `roofline` object; the reference has no profiler hooks) --------------------
 * names: "" = off, "*" = every kernel, else comma-separated kernel base names ("k_hash_fwd,k_field_bwd").  While enabled, every launch of a named kernel is
 * bracketed by an event pair on the stream it is launched on.  ngp_prof_read(index, ...) waits for and returns the durations (ms, oldest first) recorded for
 * the index-th registered kernel since the last read and writes its name; returns the count, or -1 when index is past the last registered kernel. */
int ngp_prof_enable(const char *names);
int ngp_prof_read(int index, char *name_out_host, int name_cap, float *ms_out_host, int max);

#ifdef __cplusplus
}
#endif
#endif
