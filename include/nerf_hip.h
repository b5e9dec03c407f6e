/*
 * nerf_hip.h — C ABI of libnerf_hip.so: the NeRF ray-marching render path on MI355X (gfx950).
 *
 * The upstream program (hwan0806/NeRF, a nerf-pytorch fork) has no native interface: its hot path
 * is Python calling ATen.  Each entry point below replaces the arithmetic of one upstream Python
 * function (cited per entry as file:line of the upstream tree) and is what a ctypes / cffi binding
 * on the upstream side would call.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to fp32 data unless stated; the library never allocates,
 *     frees or retains caller memory;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream); results are ordered like any other work on that stream;
 *   - return value 0 = success; < 0 = error (NERF_E_*), with a message available from
 *     nerf_last_error() on the calling thread; argument errors are detected before any launch;
 *   - tensors are dense row-major with the shapes given; "rays" is the packed per-ray record of
 *     upstream render(): [o(3) d(3) near far viewdir(3)] with a caller-given row stride in floats;
 *   - thread-compatible: the error string is thread-local; the only other state the library keeps is per-device
 *     set-up it makes on first use and guards with a mutex (which kernels have been given their LDS limit, the
 *     compute-unit count, the device address of nerf_mlp_bwd_bf16x3_static's built-in workspace);
 *   - graph capture: an entry point that has run once on a device makes no runtime call besides hipGetDevice, stream
 *     operations and kernel launches when it runs again, so make one eager call per entry point and device before
 *     capturing (the first call sets LDS limits with hipFuncSetAttribute and may load code lazily).
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_E_ARG (-1)     /* invalid argument (null pointer, bad size, unsupported shape) */
#define NERF_E_LAUNCH (-2)  /* HIP reported an error at launch */
#define NERF_E_ARCH (-3)    /* the current device is not gfx950 */

/* ---- probes ------------------------------------------------------------------------------- */
int nerf_version(void);                 /* ABI version, currently 1 */
const char* nerf_arch(void);            /* "gfx950": the only code object in the library */
const char* nerf_last_error(void);      /* message of the last failing call on this thread */
int nerf_check_device(void);            /* 0 if the current HIP device can run the kernels */

/* ---- ray generation ----------------------------------------------------------------------- */
/* get_rays (run_nerf_helpers.py:169-196).  Pixel (row j, col i) -> ray, integer pixel centres.
 * c2w: device, 12 floats, row-major 3x4.  rays_o, rays_d: [H*W,3].  intr = fx, fy, cx, cy as the
 * fp32 values of K[0][0], K[1][1], K[0][2], K[1][2]. */
int nerf_get_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w,
                  float* rays_o, float* rays_d, void* stream);

/* The random ray batch of one optimisation step taken straight from the camera, with its target colours:
 * replaces get_rays + the coords meshgrid / central crop / np.random.choice gather of the reference's training
 * loop (run_nerf.py:833-854).  pick[k] in [0, win_rows*win_cols) numbers the pixels of a window of the image
 * row-major (whole image: row0 = col0 = 0, win = H x W); pixels outside [0,H)x[0,W) are the caller's error.
 * image (optional): [H*W, 3] colours of the same view -> target [n, 3]. */
int nerf_rays_at(int64_t n, const int64_t* pick, int H, int W, int row0, int col0, int win_rows, int win_cols, float fx,
                 float fy, float cx, float cy, const float* c2w, const float* image, float* rays_o, float* rays_d,
                 float* target, void* stream);

/* nerf_rays_at with the n DISTINCT pixels of the window drawn inside the kernel: pixel k is P(k) for a permutation P of
 * [0, win_rows*win_cols) keyed by `seed` (a four-round Feistel network walked back into the range), so one launch
 * replaces the reference's np.random.choice(..., replace=False) + gather (run_nerf.py:848-851); the choice is
 * pseudo-random and reproducible from the seed, it is not numpy's stream.  pick_out (optional, [n]): the pixels drawn,
 * numbered like nerf_rays_at's pick.  n must not exceed the window's pixel count. */
int nerf_rays_random(int64_t n, uint64_t seed, int H, int W, int row0, int col0, int win_rows, int win_cols, float fx,
                     float fy, float cx, float cy, const float* c2w, const float* image, float* rays_o, float* rays_d,
                     float* target, int64_t* pick_out, void* stream);

/* ndc_rays (run_nerf_helpers.py:230-255) on n rays; sx = -1/(W/(2 focal)), sy = -1/(H/(2 focal))
 * evaluated in double by the caller and rounded to fp32, as the upstream scalar arithmetic does.
 * In-place use (o_out == o_in, d_out == d_in) is allowed. */
int nerf_ndc_rays(int64_t n, float sx, float sy, float near, const float* o_in, const float* d_in,
                  float* o_out, float* d_out, void* stream);

/* The ray record built by render() (run_nerf.py:120-149): unit view directions taken from
 * viewdir_src [n,3] (NULL = no view directions, 8-column record), optional NDC warp of (o,d) with
 * near plane ndc_near, then [o d near far viewdir].  near_v / far_v: optional per-ray bounds [n]
 * overriding the scalars. */
int nerf_pack_rays(int64_t n, const float* rays_o, const float* rays_d, const float* viewdir_src, int ndc,
                   float sx, float sy, float ndc_near, float near, float far, const float* near_v,
                   const float* far_v, float* rays, int ray_stride, void* stream);

/* ---- stratified depths + sample positions ------------------------------------------------- */
/* render_rays (run_nerf.py:419-453).  rays: [n, ray_stride] packed records.  t_vals: [S] device
 * (the caller's linspace(0,1,S)).  t_rand: [n,S] uniforms or NULL (perturb == 0).  Outputs
 * z: [n,S], pts: [n,S,3] = o + d*z. */
int nerf_sample_coarse(int64_t n, int S, const float* rays, int ray_stride, const float* t_vals,
                       const float* t_rand, int lindisp, float* z, float* pts, void* stream);

/* ---- positional encoding (stand-alone) ---------------------------------------------------- */
/* Embedder.embed (run_nerf_helpers.py:13-66): x [n,3] -> out [n, 3+6L], per frequency sin then cos. */
int nerf_encode(int64_t n, int L, const float* x, float* out, void* stream);

/* Embedder.embed for any keyword setting of the reference's Embedder whose periodic_fns are torch.sin / torch.cos in any
 * order and number (run_nerf_helpers.py:13-47: include_input, input_dims = d, num_freqs, log or linear band spacing):
 * out [n, (include_input ? d : 0) + n_freq * n_fn * d] = [x] ++ [fn_j(x * freqs[b])] band-major, function-minor.
 * freqs: DEVICE, the fp32 band values as the reference computes them; bit j of cos_mask set = function j is the cosine.
 * (nerf_encode above is the default setting, the one fused into the MLP kernels.) */
int nerf_encode_general(int64_t n, int d, int include_input, int n_freq, const float* freqs, int n_fn, unsigned cos_mask,
                        const float* x, float* out, void* stream);

/* ---- hierarchical sampling ---------------------------------------------------------------- */
/* sample_pdf (run_nerf_helpers.py:260-303) + the merge/sort and z_std of render_rays
 * (run_nerf.py:465-470,485) in one kernel.  z: [n,Sc] coarse depths (ascending);
 * weights: [n,Sc] compositing weights of the coarse pass (the kernel uses columns 1..Sc-2);
 * u: [n,Si] uniforms, or [Si] shared by all rays when u_shared != 0 (the deterministic linspace).
 * Outputs (any may be NULL except z_out): z_out [n,Sc+Si] sorted; pts_out [n,Sc+Si,3];
 * z_std [n]; samples [n,Si] (unsorted inverse-CDF samples); inds [n,Si] int64 searchsorted(right)
 * indices.  rays/ray_stride as above (only needed when pts_out != NULL).  Sc >= 3. */
int nerf_sample_fine(int64_t n, int Sc, int Si, const float* z, const float* weights, const float* u,
                     int u_shared, const float* rays, int ray_stride, float* z_out, float* pts_out,
                     float* z_std, float* samples, int64_t* inds, void* stream);

/* sample_pdf alone (run_nerf_helpers.py:260-303): bins [n,n_bins], weights [n,n_bins-1], u as
 * above -> samples [n,Si] and, if inds != NULL, the int64 searchsorted(right=True) indices. */
int nerf_sample_pdf(int64_t n, int n_bins, int Si, const float* bins, const float* weights, const float* u,
                    int u_shared, float* samples, int64_t* inds, void* stream);

/* ---- alpha compositing -------------------------------------------------------------------- */
/* raw2outputs (run_nerf.py:292-362).  raw: [n,S,4]; z: [n,S]; rays_d: direction of ray i at
 * rays_d + i*dir_stride (3 floats; dir_stride = 3 for a dense [n,3] array, or the record stride
 * when pointing at column 3 of the packed rays); noise: [n,S] already scaled by raw_noise_std, or
 * NULL.  Outputs: rgb [n,3], disp [n], acc [n], weights [n,S], depth [n]. */
int nerf_composite_fwd(int64_t n, int S, const float* raw, const float* z, const float* rays_d,
                       int dir_stride, const float* noise, int white_bkgd, float* rgb, float* disp,
                       float* acc, float* weights, float* depth, void* stream);

/* Backward of the above w.r.t. raw (what autograd derives upstream for run_nerf.py:292-362).
 * weights: the forward's output.  Upstream gradients: g_rgb [n,3], g_disp [n], g_acc [n],
 * g_weights [n,S], g_depth [n]; each may be NULL (= zero).  Output d_raw [n,S,4]. */
int nerf_composite_bwd(int64_t n, int S, const float* raw, const float* z, const float* rays_d,
                       int dir_stride, const float* noise, int white_bkgd, const float* weights,
                       const float* g_rgb, const float* g_disp, const float* g_acc,
                       const float* g_weights, const float* g_depth, float* d_raw, void* stream);

/* ---- the MLP ------------------------------------------------------------------------------ */
/* The network is NeRF(D=8, W=256, skips=[4]) (run_nerf_helpers.py:73-133) with L=10 / L=4
 * encodings fused in (run_nerf.py:40-66).  Weights are consumed from packed streams built from the flat concatenation
 * of the module's parameters in state_dict order.  One forward and one backward per arithmetic:
 *   two fp16 pieces per operand on the fp16 matrix instruction (fp32-accurate, the default of rendering and training):
 *                                              nerf_mlp_fwd_f16x2 / nerf_mlp_bwd_data_f16x2 + nerf_mlp_bwd_weights_f16x2
 *   three bf16 pieces per operand on the bf16 matrix instruction (fp32-accurate whatever the magnitudes: the fall-back
 *   of the two-piece kernels):                                                   nerf_mlp_fwd_bf16x3 / nerf_mlp_bwd_bf16x3
 *   the fp32 matrix instruction:                                                 nerf_mlp_fwd_x4     / nerf_mlp_bwd_x4
 *   bf16 operands, fp32 sums (inference only, reduced precision):                nerf_mlp_fwd_bf16
 *   already-encoded inputs, as NeRF.forward(x) receives them upstream:           nerf_mlp_fwd_encoded */
int64_t nerf_mlp_n_params(int use_viewdirs, int output_ch);   /* floats in the flat parameter vector */
int64_t nerf_mlp_packed_floats(int use_viewdirs);             /* floats in the stream of nerf_mlp_fwd_encoded */
int64_t nerf_mlp_act_floats(int use_viewdirs, int64_t n_pts); /* floats of saved activations for n_pts points */
int64_t nerf_mlp_scratch_floats(int use_viewdirs, int64_t n_pts); /* floats of backward workspace for n_pts points */

/* Where float `idx` of the nerf_mlp_fwd_encoded stream comes from: an index into the flat parameter vector, or -1
 * for a zero.  Host-only (layout introspection for tests and tools); the _table form fills a HOST array of the
 * stream's length. */
int64_t nerf_pack_source(int use_viewdirs, int output_ch, int64_t idx);
int nerf_pack_source_table(int use_viewdirs, int output_ch, int64_t* table, int64_t n);

/* flat params -> the stream of nerf_mlp_fwd_encoded. */
int nerf_pack_weights(int use_viewdirs, int output_ch, const float* params, float* packed_fwd, void* stream);

/* out[i] = table[i] < 0 ? 0 : params[table[i]] for i < n (all DEVICE): any packed stream as a gather through a map
 * the caller keeps (e.g. obtained once by packing the vector 1, 2, 3, ... and subtracting one), a few microseconds
 * after every optimiser step.  Part of the weight hand-over of NeRF.forward (run_nerf_helpers.py:109-133); upstream has
 * no counterpart, its GEMMs read the module's tensors directly. */
int nerf_gather(const int32_t* table, const float* params, float* out, int64_t n, void* stream);

/* NeRF.forward exactly as upstream calls it (run_nerf_helpers.py:109-133): x [n_pts, 63 (+27)] holds
 * already-encoded inputs (positions, then directions when the network has a view branch); out [n_pts, out_ch]
 * (out_ch = 4 with view directions, else output_ch).  act: NULL for inference; for training a buffer of
 * nerf_mlp_act_floats() floats receiving the per-layer activations and ReLU mask words the backward passes read. */
int nerf_mlp_fwd_encoded(int64_t n_pts, const float* x, int use_viewdirs, int output_ch,
                         const float* packed_fwd, float* out, float* act, void* stream);

/* Fused encode + NeRF.forward on the fp32 matrix instruction (v_mfma_f32_16x16x4_f32, 16-point tiles, two waves per
 * SIMD): run_nerf_helpers.py:109-133 behind run_nerf.py:40-66.  pts [n_pts,3]; viewdirs [n_rays,3] (ray of point p =
 * p / pts_per_ray) or NULL when the network has no view branch; out [n_pts, out_ch]; `act` as in
 * nerf_mlp_fwd_encoded.  packed_x4: nerf_mlp_packed_x4_floats() floats built by nerf_pack_weights_x4
 * from the flat parameter vector (the module's parameters concatenated in state_dict order). */
int64_t nerf_mlp_packed_x4_floats(int use_viewdirs);
int nerf_pack_weights_x4(int use_viewdirs, int output_ch, const float* params, float* packed_x4, void* stream);
int nerf_mlp_fwd_x4(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs,
                    int use_viewdirs, int output_ch, const float* packed_x4, float* out, float* act, void* stream);

/* bf16-operand / fp32-accumulate inference variant of nerf_mlp_fwd_x4 (same arguments and outputs, no `act`).
 * packed_bf16: nerf_mlp_packed_bf16_bytes() bytes built by nerf_pack_weights_bf16 from the flat
 * parameter vector (the module's parameters concatenated in state_dict order).
 * Every product's inputs are rounded to bf16 (8 significant bits), sums and heads are fp32: raw outputs
 * agree with the fp32 path to about 1e-2 of their scale. */
int64_t nerf_mlp_packed_bf16_bytes(int use_viewdirs);
int nerf_pack_weights_bf16(int use_viewdirs, int output_ch, const float* params, void* packed_bf16, void* stream);
int nerf_mlp_fwd_bf16(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs,
                      int use_viewdirs, int output_ch, const void* packed_bf16, float* out, void* stream);

/* fp32-accurate variant of nerf_mlp_fwd_x4 on the bf16 matrix instruction (same arguments and outputs; the default
 * of earlier versions, now what nerf_mlp_fwd_f16x2 hands its marked tiles to): every fp32 operand of magnitude below 3.39e38 (the largest bf16) is split exactly
 * into three bf16 pieces and each product is rebuilt from its six largest
 * piece products in fp32 accumulators, which measures the same error against fp64 as the fp32 matrix instruction
 * (NeRF.forward, run_nerf_helpers.py:109-133, behind run_nerf.py:40-66).  Non-finite values: an infinite or NaN operand
 * splits into pieces that contain a NaN, so the result is NaN (upstream: NaN as well as soon as the infinity meets
 * weights of both signs, which the next layer's sum does); a finite operand of magnitude 3.39e38 or more also gives
 * NaN where the fp32 instruction stays finite.  The ReLU of all MLP kernels keeps a NaN a NaN, like torch.relu.
 * packed_bf16x3:
 * nerf_mlp_packed_bf16x3_bytes() bytes built by nerf_pack_weights_bf16x3 from the flat parameter vector. */
int64_t nerf_mlp_packed_bf16x3_bytes(int use_viewdirs);
int nerf_pack_weights_bf16x3(int use_viewdirs, int output_ch, const float* params, void* packed_bf16x3, void* stream);
int nerf_mlp_fwd_bf16x3(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs,
                        int use_viewdirs, int output_ch, const void* packed_bf16x3, float* out, float* act, void* stream);

/* fp32-accurate forward on the fp16 matrix instruction, two pieces per operand (same arguments and outputs as
 * nerf_mlp_fwd_bf16x3; NeRF.forward, run_nerf_helpers.py:109-133, behind run_nerf.py:40-66): x ~ hi + lo * 2^-11 with
 * hi = fp16(x), lo = fp16((x - hi) * 2^11), a product = hi.hi + (hi.lo + lo.hi) * 2^-11 in two fp32 accumulators -- three
 * matrix instructions per product where the bf16 scheme issues six.  22 of an fp32's 24 significand bits are kept, which
 * stays below the rounding of the fp32 accumulation as long as the leading piece is a normal fp16; the kernel checks
 * that per layer and point (largest magnitude in [2^-11, 2^15)) and, per layer, for the weights, and writes one word per
 * tile of nerf_mlp_f16x2_tile_points() points into tile_flags (DEVICE, ceil(n_pts / tile points) words, caller-owned,
 * need not be initialised): 0 = the tile's outputs stand, 1 = the range was left.  With packed_bf16x3 (the stream of
 * nerf_pack_weights_bf16x3) the marked tiles are then recomputed by the three-piece kernel in a second launch on the
 * same stream, so out (and act) hold fp32-accurate results whatever the magnitudes, non-finite classes as documented
 * for nerf_mlp_fwd_bf16x3; with packed_bf16x3 = NULL only the first launch runs and the caller reads the flags.
 * packed_f16x2: nerf_mlp_packed_f16x2_bytes() bytes built by nerf_pack_weights_f16x2 from the flat parameter vector
 * (1,160 or 960 pairs of 1 KiB A-operand blocks hi | lo, the fp32 tables, 16 range words). */
int64_t nerf_mlp_packed_f16x2_bytes(int use_viewdirs);
int64_t nerf_mlp_f16x2_tile_points(void);
int nerf_pack_weights_f16x2(int use_viewdirs, int output_ch, const float* params, void* packed_f16x2, void* stream);
/* The same stream as one gather through an int32 map the caller keeps (DEVICE; as nerf_gather_split3): `table` holds 512
 * entries per pair block, then one per fp32 table value, then 512 per pair block of a second stream (pairs1 blocks; the
 * transposed stream of nerf_mlp_bwd_data_f16x2, or 0) -- the index into params of the value to split (or copy), or
 * -1 for a zero; `out` receives pairs x 2 KiB, n_aux floats (a multiple of 4), the 16 range words (over the first
 * stream's weights) and pairs1 x 2 KiB.  With the map taken once per model from nerf_pack_weights_f16x2 (and
 * nerf_pack_weights_bwd_f16x2) run on the vector 1/64, 2/64, 3/64, ... (exact in two fp16 pieces), the streams are
 * rebuilt in a few microseconds after every optimiser step.  Part of the weight hand-over of NeRF.forward
 * (run_nerf_helpers.py:109-133); upstream has no counterpart. */
int nerf_gather_split2(const int32_t* table, const float* params, void* out, int64_t pairs, int64_t n_aux, int64_t pairs1,
                       void* stream);
int nerf_mlp_fwd_f16x2(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs,
                       int use_viewdirs, int output_ch, const void* packed_f16x2, const void* packed_bf16x3,
                       float* out, float* act, uint32_t* tile_flags, void* stream);

/* Backward of the forward entries w.r.t. the parameters (inputs carry no gradient upstream: run_nerf.py:467 detaches
 * the samples): what autograd derives for run_nerf_helpers.py:109-133.  d_out [n_pts,out_ch]; act from any forward
 * above; scratch: nerf_mlp_scratch_floats() floats of workspace; d_params: flat gradient, same layout as the flat
 * parameter vector, OVERWRITTEN.  Fp32 matrix instruction throughout: backward-data on 16-point tiles (two waves per
 * SIMD, the mirror of nerf_mlp_fwd_x4), then dW = dY X^T on v_mfma_f32_32x32x2_f32.  packed_x4: the stream of
 * nerf_pack_weights_x4; packed_bwd_x4: nerf_mlp_packed_bwd_x4_floats() floats built by nerf_pack_weights_bwd_x4. */
int64_t nerf_mlp_packed_bwd_x4_floats(int use_viewdirs);
int nerf_pack_weights_bwd_x4(int use_viewdirs, int output_ch, const float* params, float* packed_bwd_x4, void* stream);
int nerf_mlp_bwd_x4(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                    const float* packed_x4, const float* packed_bwd_x4, float* scratch, float* d_params, void* stream);

/* The same backward on three-way bf16 splits (the mirror of nerf_mlp_fwd_bf16x3: dY tiles split
 * exactly into three bf16 pieces, W^T as split triples, six piece products, fp32 sums, in both stages).  Same d_out /
 * act / scratch / d_params as nerf_mlp_bwd_x4.  packed_bf16x3: the
 * stream of nerf_pack_weights_bf16x3 (its fp32 tables are read); packed_bwd_bf16x3: nerf_mlp_packed_bwd_bf16x3_bytes()
 * bytes built by nerf_pack_weights_bwd_bf16x3. */
int64_t nerf_mlp_packed_bwd_bf16x3_bytes(int use_viewdirs);
int nerf_pack_weights_bwd_bf16x3(int use_viewdirs, int output_ch, const float* params, void* packed_bwd_bf16x3, void* stream);
int nerf_mlp_bwd_bf16x3(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                        const void* packed_bf16x3, const void* packed_bwd_bf16x3, float* scratch, float* d_params, void* stream);

/* Both split streams in one launch, as a gather through an int32 map the caller keeps (all DEVICE).  `out` receives
 * triples0 triples (3 KiB each: 512 bf16 hi | 512 mid | 512 lo), then n_aux fp32 values (a multiple of 4), then
 * triples1 triples; `table` holds, in that order, 512 entries per triple and one per fp32 value: the index into
 * params of the value to split (or copy), or -1 for a zero.  With the map obtained once per model by running
 * nerf_pack_weights_bf16x3 and nerf_pack_weights_bwd_bf16x3 on the vector 1, 2, 3, ... and summing the pieces,
 * out = the first stream followed by the second, rebuilt in a few microseconds after every optimiser step.
 * nerf_mlp_bf16x3_triples(): triples in the forward stream (the rest of its bytes are the fp32 tables).  Part of the
 * weight hand-over of NeRF.forward (run_nerf_helpers.py:109-133); upstream has no counterpart. */
int64_t nerf_mlp_bf16x3_triples(int use_viewdirs);
int nerf_gather_split3(const int32_t* table, const float* params, void* out, int64_t triples0, int64_t n_aux,
                       int64_t triples1, void* stream);
/* nerf_gather_split2 (out2) and nerf_gather_split3 (out3) of one network in ONE launch: all four weight streams a training
 * step on the default kernels reads.  Arguments as in the two calls. */
int nerf_gather_split23(const int32_t* table2, const int32_t* table3, const float* params, void* out2, void* out3, int64_t pairs,
                        int64_t n_aux2, int64_t pairs1, int64_t triples0, int64_t n_aux3, int64_t triples1, void* stream);

/* The two stages of nerf_mlp_bwd_bf16x3 as separate calls (same arguments; the stages communicate through `scratch`
 * only, so a caller can time them, or overlap other work between them): _data fills the dY rows of every layer
 * (what autograd calls grad of the pre-activations, run_nerf_helpers.py:109-133 backwards), _weights turns them and the
 * saved activations into d_params = dY X^T per layer (split operands as well: both operands split exactly into three
 * bf16 pieces while they are staged, six piece products, fp32 sums, partial sums per slice of the point range added in
 * fixed order). */
int nerf_mlp_bwd_data_bf16x3(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                             const void* packed_bf16x3, const void* packed_bwd_bf16x3, float* scratch, void* stream);
int nerf_mlp_bwd_weights_bf16x3(int64_t n_pts, const float* act, int use_viewdirs, int output_ch, float* scratch,
                                float* d_params, void* stream);

/* Backward-data on the fp16 matrix instruction, two pieces per operand: the first stage of the parameter-gradient
 * pass (same d_out / act / scratch and the same dY rows as nerf_mlp_bwd_data_bf16x3; autograd of
 * run_nerf_helpers.py:109-133), the mirror of nerf_mlp_fwd_f16x2, three matrix instructions per product.  Gradients lie
 * far below fp16's normal range, so every point carries a power-of-two scale (the stage is linear in each point's
 * d_out and a multiplication by 2^k is exact): the point's largest |d_out| starts at 2^8 and the scale is moved after
 * every layer so that the largest value just produced sits at 2^8 again; what is stored is unscaled fp32.  The range
 * rule of nerf_mlp_fwd_f16x2 then holds per point and layer, and tile_flags (DEVICE, one word per 128-point tile of the
 * padded point range, caller-owned, need not be initialised) receives 1 where it was left (a layer that grows a
 * point's gradient by 2^7 or shrinks its largest entry by 2^19, an infinite or NaN gradient, weights out of range).
 * With the two three-piece streams (nerf_pack_weights_bf16x3 / nerf_pack_weights_bwd_bf16x3) the marked tiles are
 * recomputed by the three-piece kernel in a second launch on the same stream; with both NULL only the first launch runs.
 * packed_f16x2: the forward stream of nerf_pack_weights_f16x2 (its fp32 tables and range words are read);
 * packed_bwd_f16x2: nerf_mlp_packed_bwd_f16x2_bytes() bytes built by nerf_pack_weights_bwd_f16x2 (W^T of the layers as
 * pairs of 1 KiB A-operand blocks hi | lo), or the second stream of nerf_gather_split2.
 * tile_max (DEVICE, NERF_DY_STAGES words per tile, caller-owned, need not be initialised; or NULL): per tile and group of dY
 * rows, the largest magnitude stored, as exponent bits + 1 (0 = all zeros) -- what nerf_mlp_bwd_weights_f16x2 takes its
 * scales from.  The weight-gradient stage that follows is nerf_mlp_bwd_weights_f16x2 or nerf_mlp_bwd_weights_bf16x3. */
#define NERF_DY_STAGES 18
#define NERF_DW_CTL_WORDS 4096
int64_t nerf_mlp_packed_bwd_f16x2_bytes(int use_viewdirs);
int nerf_pack_weights_bwd_f16x2(int use_viewdirs, int output_ch, const float* params, void* packed_bwd_f16x2, void* stream);
int nerf_mlp_bwd_data_f16x2(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                            const void* packed_f16x2, const void* packed_bwd_f16x2, const void* packed_bf16x3,
                            const void* packed_bwd_bf16x3, float* scratch, uint32_t* tile_flags, uint32_t* tile_max,
                            void* stream);

/* The weight-gradient stage on the fp16 matrix instruction, three matrix instructions per product into one accumulator
 * (same act / scratch / d_params as nerf_mlp_bwd_weights_bf16x3; dW = dY X^T of autograd, run_nerf_helpers.py:109-133
 * backwards).  dY is multiplied by one power of two per group of rows and batch (from tile_max: the group's largest
 * magnitude goes to [2^13, 2^14)) and split into fp16(x), fp16(x - hi) and fp16(x) * 2^-11; the activations into
 * fp16(x) and fp16((x - hi) * 2^11); product = hi.hi + residual.hi + (hi * 2^-11).lo.  Every activation keeps 22 bits of
 * its own magnitude; the gradient side is exact to 2^-39 of the group's largest gradient in the batch (full relative
 * precision over 17 octaves below it).  Error of a product against fp64 below an fp32 accumulation's of the same sum,
 * relative to the product's largest entry; a unit whose gradients stay a million times below its layer's largest for
 * the whole batch keeps about five digits (nerf_mlp_bwd_weights_bf16x3 keeps all of them).  If any word of fwd_flags (the tile flags of
 * nerf_mlp_fwd_f16x2 for these points, or NULL) or bwd_flags (those of nerf_mlp_bwd_data_f16x2) is set, the three-piece
 * kernel computes the whole batch instead; the choice is made on the device (both kernels are launched, one returns at
 * once), nothing synchronises.  ctl: NERF_DW_CTL_WORDS words of DEVICE workspace. */
int nerf_mlp_bwd_weights_f16x2(int64_t n_pts, const float* act, int use_viewdirs, int output_ch, float* scratch,
                               float* d_params, const uint32_t* fwd_flags, const uint32_t* bwd_flags, const uint32_t* tile_max,
                               uint32_t* ctl, void* stream);

/* torch.optim.Adam's update (what upstream's optimizer.step() runs: run_nerf.py:217-218, 868-881) over up to
 * nerf_adam_max_runs() runs of parameters in ONE launch.  Run i: counts[i] fp32 elements at params[i], their gradients at
 * grads[i] and both moments at exp_avg[i] / exp_avg_sq[i] (all DEVICE; the four arrays of pointers and counts are HOST
 * arrays), updated in place: g += weight_decay * p (if not 0); m = beta1 m + (1 - beta1) g; v = beta2 v + (1 - beta2) g g;
 * p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps), products and sums in fp64 from the fp32 operands,
 * every assignment rounded to fp32, as the library optimiser's fused kernel computes them.  t, the count of this update
 * (1 for the first): step_dev[i] points to run i's count as a DEVICE float (the caller increments it beforehand; a graph
 * replay then reads the current value), or step_dev = NULL (or an entry NULL) and `step` is used.  lr_dev: the learning
 * rate as a DEVICE float, or NULL and `lr` is used.  amsgrad / maximize are not covered. */
int nerf_adam_max_runs(void);
int nerf_adam_step(int n_runs, const int64_t* counts, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const float* const* step_dev, float step, const float* lr_dev, double lr,
                   double beta1, double beta2, double eps, double weight_decay, void* stream);

/* The photometric loss of a batch and its gradient (upstream: img2mse = mean((x - y)^2), run_nerf_helpers.py:8, called
 * on the fine and the coarse colours at run_nerf.py:863-869; the gradient is autograd's of that expression).  x, y: n
 * DEVICE floats, 1 <= n <= nerf_mse_max_elems() (a training batch; one workgroup reduces it in fp64, in a fixed order);
 * out: one DEVICE float.  nerf_mse_bwd: dx[i] = grad_out[0] * (2 / n) * (x[i] - y[i]), grad_out one DEVICE float. */
int64_t nerf_mse_max_elems(void);
int nerf_mse_fwd(int64_t n, const float* x, const float* y, float* out, void* stream);
int nerf_mse_bwd(int64_t n, const float* x, const float* y, const float* grad_out, float* dx, void* stream);
/* Both losses of a step against one target in one launch (run_nerf.py:863-870: img_loss = img2mse(rgb, target_s),
 * img_loss0 = img2mse(extras['rgb0'], target_s), loss = img_loss + img_loss0): out0, out1, out_sum one DEVICE float each,
 * out_sum = out0 + out1 added in fp32.  nerf_mse_pair_bwd: d_k[i] = (grad_k[0] + grad_sum[0]) * (2 / n) * (x_k[i] - y[i]);
 * each of grad0, grad1, grad_sum is one DEVICE float or NULL (no gradient arrived for that output). */
int nerf_mse_pair_fwd(int64_t n, const float* x0, const float* x1, const float* y, float* out0, float* out1, float* out_sum,
                      void* stream);
int nerf_mse_pair_bwd(int64_t n, const float* x0, const float* x1, const float* y, const float* grad0, const float* grad1,
                      const float* grad_sum, float* d0, float* d1, void* stream);

/* ---- nn.Linear for network shapes the fused kernels do not cover --------------------------------------------------
 * (other --netdepth / --netwidth / --multires, --i_embed -1: NeRF.forward layer by layer, run_nerf_helpers.py:109-133,
 * and autograd of it), fp32 on the fp32 matrix instruction; all pointers DEVICE, row-major, ld* = row strides in floats.
 * nerf_linear_fwd:         y [n, n_out] = act(x [n, n_in] w^T + b), w [n_out, n_in] as nn.Linear stores it; b may be
 *                          NULL; relu != 0 applies max(., 0) (a NaN stays a NaN).
 * nerf_relu_grad:          out[i] = y[i] > 0 ? dy[i] : 0 over n values (the gradient entering a ReLU layer's products).
 * nerf_linear_bwd_data:    dx [n, n_in] = dy [n, n_out] w.
 * nerf_linear_bwd_weights: dw [n_out, n_in] = dy^T x and db [n_out] = column sums of dy (db may be NULL), summed over
 *                          slices of the points in fixed order (bitwise reproducible); workspace:
 *                          nerf_linear_workspace_floats(n_in, n_out) floats, contents irrelevant. */
int nerf_linear_fwd(int64_t n, int n_in, int n_out, const float* x, int64_t ldx, const float* w, const float* b, int relu,
                    float* y, int64_t ldy, void* stream);
int nerf_relu_grad(int64_t n, const float* dy, const float* y, float* out, void* stream);
int nerf_linear_bwd_data(int64_t n, int n_in, int n_out, const float* dy, int64_t ldy, const float* w, float* dx, int64_t ldx,
                         void* stream);
int64_t nerf_linear_workspace_floats(int n_in, int n_out);
int nerf_linear_bwd_weights(int64_t n, int n_in, int n_out, const float* dy, int64_t ldy, const float* x, int64_t ldx, float* dw,
                            float* db, float* workspace, void* stream);

/* nerf_mlp_bwd_bf16x3 for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): no counterpart upstream.  Same arguments and results, bit for bit, minus
 * the workspace: the dY rows and partial-gradient slabs live in an array the library's code object holds on every
 * device it is loaded on (about 40 MB per device until the process ends).  Limits: n_pts <= NERF_STATIC_BWD_MAX_PTS
 * per call, otherwise NERF_E_ARG; the call allocates nothing; every call on one device shares that device's array,
 * so calls that overlap on one device (two streams, two threads) race -- serialise them.  The FIRST call on a device
 * looks up the array's device address (hipGetSymbolAddress, which may load the code object and synchronise) and sets
 * LDS limits: make it outside graph capture; later calls only enqueue. */
#define NERF_STATIC_BWD_MAX_PTS 2048
int64_t nerf_mlp_bwd_static_max_points(void);
int nerf_mlp_bwd_bf16x3_static(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                               const void* packed_bf16x3, const void* packed_bwd_bf16x3, float* d_params, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */
