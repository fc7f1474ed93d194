/*
 * rk.h - C ABI of librk_hip.so: the MI355X (gfx950) hot path of a randomized-
 * preconditioned Krylov engine, as a drop-in behind rlaopt's LinOp / Preconditioner /
 * Solver protocol (see INTEGRATION.md for the reference-side ctypes binding).
 *
 * Conventions
 *   - All matrix pointers are DEVICE pointers (e.g. torch `tensor.data_ptr()`), row-major,
 *     with explicit leading dimensions counted in ELEMENTS.  No torch types cross the ABI.
 *   - `dtype`: RK_F32 or RK_F64 selects the arithmetic type of every matrix argument
 *     unless a parameter says otherwise.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - Every entry point returns 0 on success and a negative rk_status otherwise; it never
 *     throws and never synchronises the device.  `rk_last_error()` returns a thread-local
 *     message for the last failure on the calling thread.
 *   - The library owns no user memory.  Kernels that need scratch take a caller-provided
 *     `ws` buffer whose size is returned by the matching `*_ws_bytes` query.
 *
 * Reference citations are paths under the upstream repository (udellgroup/rlaopt @
 * 2025-05-16); SURVEY.md section 8 row ids in brackets.
 */
#ifndef RK_H_
#define RK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { RK_F32 = 0, RK_F64 = 1 } rk_dtype;

typedef enum {
    RK_RBF = 0,      /* exp(-sum(((x-y)/l)^2)/2)                rlaopt/kernels/standard.py:46-52 */
    RK_LAPLACE = 1,  /* exp(-sum(|x-y|/l))                       rlaopt/kernels/standard.py:55-61 */
    RK_MATERN12 = 2, /* exp(-D), D = sqrt(sum(((x-y)/l)^2))      rlaopt/kernels/standard.py:64-69 */
    RK_MATERN32 = 3, /* (1+sqrt3 D) exp(-sqrt3 D)                rlaopt/kernels/standard.py:72-77 */
    RK_MATERN52 = 4  /* (1+sqrt5 D+5/3 D^2) exp(-sqrt5 D)        rlaopt/kernels/standard.py:80-85 */
} rk_kernel_kind;

typedef enum {
    RK_OK = 0,
    RK_ERR_INVALID = -1, /* bad argument (null pointer, size, alignment, unsupported k)  */
    RK_ERR_LAUNCH = -2,  /* HIP reported an error at launch                               */
    RK_ERR_WORKSPACE = -3 /* caller workspace too small                                   */
} rk_status;

/* Thread-local description of the last error on this thread ("" if none). */
const char* rk_last_error(void);
/* ABI version (bumped on any signature change). */
int rk_abi_version(void);

/* ------------------------------------------------------------------------------------
 * Kernel matrices  [a1-a4]
 *
 * Entry K[i][j] = scale * kappa(X1[row1(i)], X2[row2(j)]) with row1(i) = idx1 ? idx1[i] : i
 * (same for row2).  The lengthscale is `ell_scalar` unless `ell_vec` (device, length d,
 * same dtype) is non-NULL (rlaopt/kernels/configs.py:14,50-60).  idx arrays are device
 * int64 (the `blk` tensors of rlaopt/kernels/base.py:104-128).
 * ---------------------------------------------------------------------------------- */

/* Dense block K (n1 x n2, leading dim ldk) written to HBM: the "materialised shard" that
 * the HBM-bound product rk_matmat_skinny streams.  Replaces the symbolic LazyTensor of
 * rlaopt/kernels/base.py:88-101 and the per-worker cache of base.py:183-204. */
int rk_kernel_materialize(int dtype, int kind,
                          const void* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                          const void* X2, int64_t n2, int64_t ldx2, const int64_t* idx2,
                          int64_t d, double ell_scalar, const void* ell_vec, double scale,
                          void* K, int64_t ldk, void* stream);

/* Fused matrix-free product Y (n1 x k) = scale * K(X1[idx1], X2[idx2]) @ V (n2 x k),
 * 1 <= k <= 16; K is never written.  Replaces `LazyTensor @ x` of
 * rlaopt/kernels/base.py:43-47,112 and the worker-side products of
 * rlaopt/kernels/utils.py:10-56 (row / block oracles).  The transpose product is the same
 * call with (X1, idx1) and (X2, idx2) exchanged.  n1 / n2 count the LOGICAL points (the
 * idx lengths when idx is given).  If `accumulate` != 0, Y += result. */
size_t rk_kernel_matmat_ws_bytes(int dtype, int64_t n1, int64_t n2, int64_t k);
/* Symmetric operands - X1 == X2 (same pointer, same ldx), idx1 == idx2, float32, k <= 4, d <= 64, a
 * gram_center: with a workspace of at least max(rk_kernel_matmat_ws_bytes, this) bytes the product
 * evaluates only the tiles on and above the diagonal and uses each for y[i] += P v[j] and
 * y[j] += P^T v[i] (half the matrix-core and exponential work; fixed summation order). */
size_t rk_kernel_matmat_sym_ws_bytes(int dtype, int64_t n, int64_t k);
int rk_kernel_matmat(int dtype, int kind,
                     const void* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                     const void* X2, int64_t n2, int64_t ldx2, const int64_t* idx2,
                     int64_t d, double ell_scalar, const void* ell_vec, double scale,
                     const void* V, int64_t k, int64_t ldv,
                     void* Y, int64_t ldy, int accumulate,
                     const void* gram_center, void* ws, size_t ws_bytes, void* stream);

/* The same product for a caller that multiplies by the SAME X2 side again and again (SAP's row
 * oracle K(X[blk], X) of rlaopt/kernels/base.py:104-128: a new blk, the same X, every step): the
 * matrix-core formulation first converts both point sets to split bf16 images; with `x2_image`
 * (device scratch of rk_kernel_x2_image_bytes bytes that the caller keeps between calls) the X2 image
 * is written there, and a later call with `x2_image_ready` != 0 skips the conversion (96 us of a
 * 2.5 ms product at n2 = 500 000).  The caller vouches that X2, idx2, d, the lengthscale and
 * gram_center are those of the call that wrote the image and that the calls are ordered on one
 * stream.  `*x2_image_valid` (host int, may be NULL) tells whether the image holds the converted
 * points after this call (0 when another formulation ran: then never pass ready != 0).
 * x2_image == NULL is rk_kernel_matmat.  rk_kernel_x2_image_bytes returns 0 when no image would
 * be used (float64, Laplace / Matern-1/2, d > 128, symmetric operands). */
size_t rk_kernel_x2_image_bytes(int dtype, int kind, int64_t n1, int64_t n2, int64_t d, int64_t k,
                                int symmetric_operands);
int rk_kernel_matmat_cached(int dtype, int kind,
                            const void* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                            const void* X2, int64_t n2, int64_t ldx2, const int64_t* idx2,
                            int64_t d, double ell_scalar, const void* ell_vec, double scale,
                            const void* V, int64_t k, int64_t ldv,
                            void* Y, int64_t ldy, int accumulate,
                            const void* gram_center, void* ws, size_t ws_bytes,
                            void* x2_image, size_t x2_image_bytes, int x2_image_ready,
                            int* x2_image_valid, void* stream);

/* `gram_center` (device, length d, same dtype; may be NULL) opts in to the matrix-core
 * formulation of the float32 product (RBF, Matern-3/2, Matern-5/2, d <= 128), which expands
 * |x - y|^2 = |x'|^2 + |y'|^2 - 2 x'.y' with x' = (x - gram_center) / ell.  Its rounding error
 * grows with |x'|^2, so by passing a centre the caller vouches that
 *     max_i |(x_i - gram_center) / ell|^2 <= RK_GRAM_MAX_NORM2
 * over the rows of X1 and X2 (rk_kernel_gram_bound computes that maximum).  Entries then agree
 * with the direct (x - y)^2 evaluation to ~1e-5 absolute, the diagonal of K(X, X) included
 * (reference tolerance for float32: 1e-4, tests/kernels/test_standard.py).  NULL, float64,
 * Laplace, Matern-1/2 (not smooth in the squared distance) and d > 128 always take the direct-
 * difference kernels.  pykeops (the reference's engine) evaluates the direct form. */
#define RK_GRAM_MAX_NORM2 32.0

/* center[c] = mean of column c over <= 4096 evenly strided rows of X (n x d): a deterministic,
 * cheap origin for the expansion above (the kernels are translation invariant, any point near
 * the data will do). */
int rk_kernel_center(int dtype, const void* X, int64_t n, int64_t ldx, int64_t d,
                     void* center, void* stream);
/* bound[0] = max(bound[0], max_i sum_c ((X[i][c] - center[c]) / ell_c)^2); `bound` is ONE
 * device element of `dtype` that the caller zero-initialises (and may carry over several point
 * sets).  Rows containing NaN give +inf. */
int rk_kernel_gram_bound(int dtype, const void* X, int64_t n, int64_t ldx, int64_t d,
                         double ell_scalar, const void* ell_vec, const void* center,
                         void* bound, void* stream);

/* Symmetric operators K(X, X), "sympack" storage: columns are cut into strips of
 * W = 256 (f32) / 128 (f64) columns; strip s keeps rows 0 .. min(n, (s+1) W) - 1 of its W
 * columns (the part on and above its diagonal block), row-major with leading dimension W,
 * strips concatenated; a rank keeps the strips s = strip_first + t * strip_stride (cyclic over
 * ranks).  Half the bytes of the row-major block; rk_symm_matmat reads every stored entry once
 * and uses it for y[i] += K[i][j] v[j] AND y[j] += K[i][j] v[i].  The result of rk_symm_matmat
 * is this rank's PARTIAL sum over its strips (complete when strip_stride == 1); partials of
 * different ranks add up to K V (all-reduce).  1 <= k <= 4 per call (register kernel); with
 * store_dtype == dtype also 5 <= k <= 16 (matrix-core strip kernel: both uses of a stored entry
 * as v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 products out of LDS-staged tiles, one pass
 * for up to 16 right-hand sides).  No reference
 * counterpart: the reference never exploits A1 is A2 (rlaopt/kernels/base.py:23-128). */
/* `store_dtype` is the element type of the packed matrix: equal to `dtype`, or RK_F32 under
 * RK_F64 arithmetic ("fp32 storage, fp64 iteration": entries are computed in float64 and
 * rounded once; products convert them back and accumulate in float64 - fp32 HBM traffic with
 * float64 vectors, so a Krylov solve is not limited by fp32 round-off in its recurrences). */
size_t rk_sympack_elems(int store_dtype, int64_t n, int64_t strip_first, int64_t strip_stride);
int rk_kernel_materialize_sympack(int dtype, int store_dtype, int kind, const void* X, int64_t n,
                                  int64_t ldx, int64_t d, double ell_scalar, const void* ell_vec,
                                  double scale, int64_t strip_first, int64_t strip_stride,
                                  void* Kp, void* stream);
/* The same packed triangle, float32, written FROM THE MATRIX CORES: d <= 64, kernels smooth in the squared distance (rbf,
 * matern32, matern52), `center` = a centre for which |(x - c) / ell|^2 <= RK_GRAM_MAX_NORM2 holds over the rows of X
 * (rk_kernel_center + rk_kernel_gram_bound: the guard of the matrix-free product).  The points are split into two float16
 * pieces once (workspace), the dot products of a 16 x 16 block are three exact matrix-core products, the kernel function
 * runs on the accumulators: 4 x the rate of the vector-unit evaluation (the 80 GB triangle of n = 200 000 in ~20 ms: the
 * speed HBM takes the writes at), entries to the accuracy of the matrix-free product.  Replaces what the reference
 * re-evaluates in every product (rlaopt/kernels/base.py:43-44). */
size_t rk_kernel_materialize_sympack_mc_ws_bytes(int64_t n);

int rk_kernel_materialize_sympack_mc(int kind, const void* X, int64_t n, int64_t ldx, int64_t d, double ell_scalar,
                                     const void* ell_vec, const void* center, double scale, int64_t strip_first,
                                     int64_t strip_stride, void* Kp, void* ws, size_t ws_bytes, void* stream);
/* The dense row-major block of rk_kernel_materialize (float32, d <= 64, rbf / matern32 / matern52, a centre within the norm
 * bound) evaluated the same way: K(X1[idx1], X2[idx2]) - the |blk| x |blk| blocks of a sketch-and-project step
 * (rlaopt/solvers/sap.py:64-66) and row-major shards. */
size_t rk_kernel_materialize_mc_ws_bytes(int64_t n1, int64_t n2);
int rk_kernel_materialize_mc(int kind, const void* X1, int64_t n1, int64_t ldx1, const int64_t* idx1, const void* X2,
                             int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d, double ell_scalar,
                             const void* ell_vec, const void* center, double scale, void* K, int64_t ldk, void* ws,
                             size_t ws_bytes, void* stream);
size_t rk_symm_matmat_ws_bytes(int dtype, int store_dtype, int64_t n, int64_t k,
                               int64_t strip_first, int64_t strip_stride);
/* Y (n x k) = alpha * (partial) K V + diag_shift * V */
int rk_symm_matmat(int dtype, int store_dtype, const void* Kp, int64_t n, int64_t strip_first,
                   int64_t strip_stride, const void* V, int64_t k, int64_t ldv, double alpha,
                   double diag_shift, void* Y, int64_t ldy, void* ws, size_t ws_bytes,
                   void* stream);

/* The same product for 1 <= k <= 16 right-hand sides (float32 storage and vectors) on the 16-BIT matrix cores:
 * stored entries and V are split into two float16 pieces each (three exact products per pair, float32
 * accumulation; entry error <= 2.4e-7 relative), both uses of a stored tile out of one LDS image (row reads and
 * transposing reads).  Work items are blocks of four strips side by side (csrc/rk_symmblk.hip): a row's partial sum is
 * carried across the strips, so the partial tables are a third of the strip kernels' and the product streams the
 * packed bytes at 6.0 - 6.9 TB/s for every k (n = 200 000: 11.6 ms at k = 1, 13.3 ms at k = 16).  kmax: an upper bound
 * on the magnitude of the stored entries (the operator's const_scaling).  The package's default for float32 products
 * (rlaopt/solvers/pcg.py:58, one or several columns) from 24 GB of packed matrix on; bitwise reproducible. */
size_t rk_symm_matmat_f16x2_ws_bytes(int64_t n, int64_t k, int64_t strip_first, int64_t strip_stride);
int rk_symm_matmat_f16x2(const void* Kp, int64_t n, int64_t strip_first, int64_t strip_stride, double kmax,
                         const void* V, int64_t k, int64_t ldv, double alpha, double diag_shift, void* Y,
                         int64_t ldy, void* ws, size_t ws_bytes, void* stream);

/* Wide product with the packed matrix: Y (n x k) = alpha * (this rank's partial of) K V for 17 <= k <= 1024
 * columns - the Nystrom sketch product Y = K Omega (rlaopt/preconditioners/nystrom.py:79-80, the one
 * n x n x r product of a solve; the reference sends Omega through its worker queues,
 * rlaopt/linops/base.py:202-267).  float32 storage and vectors.  The strips are read in place (no row-major
 * panels); both factors are split into two float16 pieces (three exact products on the 16-bit matrix
 * cores, float32 accumulation, entry error <= 2.4e-7 relative; RK_SYMGEMM_SPLIT=bf16x3: three bfloat16
 * pieces, six products, float32-exact).  kmax: an upper bound on the stored entries' magnitude (the
 * operator's const_scaling: kernel values are <= 1).  Overwrites Y. */
size_t rk_symm_gemm_ws_bytes(int64_t n, int64_t k);
int rk_symm_gemm(const void* Kp, int64_t n, int64_t strip_first, int64_t strip_stride, double kmax,
                 const void* V, int64_t k, int64_t ldv, double alpha, void* Y, int64_t ldy, void* ws,
                 size_t ws_bytes, void* stream);
/* The same product for a V on the FLOAT16 GRID - every entry, times the power of two the kernel scales its column by,
 * exactly a float16 number; true of any matrix that was rounded to float16 once, such as a Gaussian test matrix the
 * caller draws itself (the sketch Omega of rlaopt/sketches/gauss.py:46-52 is random either way: rounding its entries to
 * 11 significant bits leaves it a Gaussian sketch).  V then has ONE piece: two exact products per pair instead of three
 * and half the bytes of V tiles through each CU - the two things rk_symm_gemm's time is made of.  Entries off the grid are
 * ROUNDED to it (the result is K round(V)); same arguments, same workspace, same reproducibility. */
int rk_symm_gemm_rhs16(const void* Kp, int64_t n, int64_t strip_first, int64_t strip_stride, double kmax,
                       const void* V, int64_t k, int64_t ldv, double alpha, void* Y, int64_t ldy, void* ws,
                       size_t ws_bytes, void* stream);

/* The same wide product for an operator that stores NOTHING (matrix-free): Y (n1 x k) = alpha * scale *
 * K(X1, X2) V for up to 1024 columns - Y = A @ Omega of rlaopt/preconditioners/nystrom.py:79 as the reference itself
 * forms it, matrix-free (rlaopt/kernels/base.py:43-44,58-59: a pykeops reduction per call).  The kernel entries are
 * evaluated tile by tile inside the product (X2 tile x X1 rows on the 16-bit matrix cores, the kernel function on the
 * accumulators, split, straight into the operand image of the main product): no n x n storage, no temporary panels.
 * float32, d <= 64, kind in {RK_RBF, RK_MATERN32, RK_MATERN52}; `center` (d floats, device) is REQUIRED and the caller
 * vouches for max |(x - center) / ell|^2 <= RK_GRAM_MAX_NORM2 over both point sets, exactly as for the matrix-core
 * form of rk_kernel_matmat (rk_kernel_center / rk_kernel_gram_bound).  X1 == X2 (same pointer, size, stride) shares
 * one image.  Overwrites Y. */
size_t rk_kernel_sketch_ws_bytes(int64_t n1, int64_t n2, int64_t k);
int rk_kernel_sketch_matmat(int kind, const void* X1, int64_t n1, int64_t ldx1, const void* X2, int64_t n2,
                            int64_t ldx2, int64_t d, double ell_scalar, const void* ell_vec, const void* center,
                            double scale, const void* V, int64_t k, int64_t ldv, double alpha, void* Y, int64_t ldy,
                            void* ws, size_t ws_bytes, void* stream);
/* The same product for a V on the float16 grid (see rk_symm_gemm_rhs16): one piece of V, two products per pair. */
int rk_kernel_sketch_matmat_rhs16(int kind, const void* X1, int64_t n1, int64_t ldx1, const void* X2, int64_t n2,
                            int64_t ldx2, int64_t d, double ell_scalar, const void* ell_vec, const void* center,
                            double scale, const void* V, int64_t k, int64_t ldv, double alpha, void* Y, int64_t ldy,
                            void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Transport  [a5, e]: RCCL behind the C ABI, one communicator per rank (one rank per GPU).
 * Replaces the reference's worker processes, pickled task / result queues and host staging
 * (rlaopt/linops/base.py:157-176,202-267): ROW-mode concat -> rk_comm_allgather, COLUMN-mode /
 * transpose sum -> rk_comm_allreduce_sum (rlaopt/linops/distributed.py:40-50,81-91), partial
 * products of cyclically dealt strips -> rk_comm_reduce_scatter_sum.  librk_hip.so does not link
 * RCCL: the library is bound at first use (the copy already mapped into the process - e.g.
 * PyTorch's - or the system's librccl.so).  rk_comm_unique_id: rank 0 fills 128 bytes, the caller
 * ships them to the other ranks by any side channel (a file, MPI, torch.distributed's store);
 * rk_comm_create is collective over the ranks.  Buffers are device pointers, `count` in elements of
 * `dtype`; all-gather / reduce-scatter take the per-rank count; every call is queued on `stream`.
 * ---------------------------------------------------------------------------------- */
int rk_comm_unique_id(void* id128);
int rk_comm_create(void** comm, int nranks, int rank, const void* id128);
int rk_comm_destroy(void* comm);
int rk_comm_allreduce_sum(void* comm, void* buf, int64_t count, int dtype, void* stream);
int rk_comm_allgather(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype, void* stream);
int rk_comm_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype,
                               void* stream);

/* Wire buffer of the reduce-scatter that sums the ranks' partial products  [a5, e]: out (slots x (step k + t)),
 * slot r = [ rows r step .. (r + 1) step of the n x k partial (zero beyond row n) | tail (t float64 values,
 * converted) ], in `dtype`.  One launch instead of a zero-fill, a strided copy and a broadcast; the transport
 * it feeds replaces the reference's result queue + host concat/sum (rlaopt/linops/distributed.py:40-50,81-91). */
int rk_spmd_pack(int dtype, const void* part, int64_t n, int64_t k, int64_t ldp, const double* tail, int64_t t,
                 int64_t step, int64_t slots, void* out, void* stream);

/* ------------------------------------------------------------------------------------
 * SAP step size  [a12]
 *
 * lambda_max(Pinv (K + reg I)) by power iteration with the stopping rule of
 * rlaopt/spectral_estimators/spectral_norm.py:11-29 (|sig_new - sig| <= rtol * sig, at most
 * max_iters steps), the whole loop queued on `stream` with NO host synchronisation: the
 * stopping test is evaluated on the device and freezes the state, so what is left in
 * state[0] is the value the reference returns.  K: the materialised block oracle K(X[blk],
 * X[blk]) (m x m, ldk; rlaopt/solvers/sap.py:89-111).  Pinv x = (x - U (M (U^T x))) / rho with
 * the Nystrom factors U (m x r) and M (r x r float64 row-major, or - mdiag != 0 - a length-r
 * float64 diagonal); U == NULL: Pinv = identity (rho must be 1).  v: start vector in, last
 * iterate out.  state: device float64[3] = {eigenvalue estimate, converged flag, iterations}.
 * Replaces ~10 launches and 2 host read-backs per iteration of the reference-shaped loop.
 * ---------------------------------------------------------------------------------- */
/* SAP / ASkotch iterate update (rlaopt/solvers/sap.py:139-175) on contiguous n x k iterates
 * (ld == k), two launches instead of a dozen element-wise / index_put kernels:
 *   accel != 0:  W <- Y;  V <- beta V + (1 - beta) Y;  on the rows blk[0 .. nb) (distinct):
 *                W -= eta D, V -= (eta gamma) D;  then Y <- alpha V + (1 - alpha) W everywhere
 *   accel == 0:  W[blk] -= eta D
 * D: nb x k block update (Pinv applied to the block gradient).  eta: `eta_dev` (ONE device
 * element of `dtype`, e.g. 1 / state[0] of rk_power_iteration - no host read-back) or, when
 * eta_dev is NULL, the host value `eta_host`. */
int rk_sap_update(int dtype, int64_t n, int64_t k, void* W, void* V, void* Y, int64_t ld,
                  const int64_t* blk, int64_t nb, const void* D, int64_t ldd,
                  const void* eta_dev, double eta_host, int accel, double beta, double gamma,
                  double alpha, void* stream);
size_t rk_power_iteration_ws_bytes(int dtype, int64_t m, int64_t r);
int rk_power_iteration(int dtype, const void* K, int64_t m, int64_t ldk, double reg,
                       const void* U, int64_t r, int64_t ldu, const double* M, int mdiag,
                       double rho, void* v, int64_t max_iters, double rtol, double* state,
                       void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * HBM-streaming skinny products  [a3 cached shard product, a6 dense tensor operator,
 * a9 Nystrom apply, a11 block dots]
 * ---------------------------------------------------------------------------------- */

/* Y (m x k) = alpha * A (m x n) @ V (n x k) + beta * C (m x k) + diag_shift * V[row0 + i]
 * for 1 <= k <= 16.  `C` may be NULL when beta == 0 and may alias Y.  The diag_shift term
 * folds "A @ P + reg * P" (rlaopt/solvers/pcg.py:58, models/linsys.py:97) into the
 * product when A is the row shard [row0, row0 + m) of a square operator.
 * The dominant, roofline-graded kernel: algorithmic bytes = (m*n + n*k + m*k)*sizeof. */
size_t rk_matmat_skinny_ws_bytes(int dtype, int64_t m, int64_t n, int64_t k);
int rk_matmat_skinny(int dtype, const void* A, int64_t m, int64_t n, int64_t lda,
                     const void* V, int64_t k, int64_t ldv,
                     double alpha, double beta, const void* C, int64_t ldc,
                     double diag_shift, int64_t row0,
                     void* Y, int64_t ldy, void* ws, size_t ws_bytes, void* stream);

/* Y (n x k) = alpha * A^T @ V + beta * C   with A (m x n) row-major, V (m x k),
 * 1 <= k <= 16.  Serves `x @ A` / `A.T @ x` on dense operators
 * (rlaopt/linops/base.py:104-111) and the tall inner products U^T R, P^T AP, R^T Z
 * (rlaopt/preconditioners/nystrom.py:114, rlaopt/solvers/pcg.py:61,79).
 * `out_f64` != 0 writes Y as float64 regardless of dtype (partials are always summed in
 * float64). */
size_t rk_matmat_skinny_t_ws_bytes(int dtype, int64_t m, int64_t n, int64_t k);
int rk_matmat_skinny_t(int dtype, const void* A, int64_t m, int64_t n, int64_t lda,
                       const void* V, int64_t k, int64_t ldv,
                       double alpha, double beta, const void* C, int64_t ldc,
                       void* Y, int64_t ldy, int out_f64,
                       void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Dense MFMA GEMMs  [a7 sketch apply, a8 Nystrom assembly]
 * ---------------------------------------------------------------------------------- */

/* C (m x n) = alpha * op(A) (m x kd) @ op(B) (kd x n) + beta * C, row-major, MFMA
 * (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64).  transX != 0 means the operand is
 * stored transposed (A stored kd x m).  Serves Y = A @ Omega (rlaopt/sketches/sketch.py:
 * 86-95), U = Y @ M and dense `A @ X` with many columns.  When the output has too few
 * 128 x 128 tiles to fill the chip the reduction dimension is split (fixed-order reduce,
 * scratch from rk_gemm_ws_bytes; 0 when no split is planned). */
size_t rk_gemm_ws_bytes(int dtype, int64_t m, int64_t n, int64_t kd);
int rk_gemm(int dtype, int transA, int transB, int64_t m, int64_t n, int64_t kd,
            double alpha, const void* A, int64_t lda, const void* B, int64_t ldb,
            double beta, void* C, int64_t ldc, void* ws, size_t ws_bytes, void* stream);

/* C (m x n, FLOAT64) = A^T @ B with A (rows x m), B (rows x n) of `dtype`, accumulated in
 * float64 on the f64 matrix cores and split over `rows` (deterministic two-stage
 * reduction).  Serves the Gram-type products Omega^T Y and Y^T Y of the Nystrom assembly
 * (rlaopt/preconditioners/nystrom.py:82,94-95). */
size_t rk_gram_tn_ws_bytes(int64_t rows, int64_t m, int64_t n);
int rk_gram_tn(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb,
               int64_t rows, int64_t m, int64_t n, double* C, int64_t ldc,
               void* ws, size_t ws_bytes, void* stream);

/* Nystrom shifted-inverse apply  [a9]:  Y (n x k) = (X - U (M (U^T X))) / rho  for the factors of
 * rlaopt/preconditioners/nystrom.py:112-132 - both reference branches reduce to this form: float64
 * M = diag(S / (S + rho)) (mdiag != 0: M is a length-r vector), float32 M = (rho diag(1 / S) + U^T U)^-1
 * (r x r row-major float64) - in three launches: U^T X (float64 accumulation), the r x r step, the rank-r
 * update fused with the scaling.  U: n x r of `dtype`; 1 <= k <= 16.  (Row-sharded U: the host composes
 * rk_matmat_skinny_t + all-reduce + rk_matmat_skinny instead.) */
size_t rk_nystrom_inv_apply_ws_bytes(int dtype, int64_t n, int64_t r, int64_t k);
int rk_nystrom_inv_apply(int dtype, const void* U, int64_t n, int64_t r, int64_t ldu, const double* M, int mdiag,
                         double rho, const void* X, int64_t k, int64_t ldx, void* Y, int64_t ldy, void* ws,
                         size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Block-PCG vector updates  [a11]   (all n x k row-major, k <= 16, coefficient
 * matrices k x k FLOAT64 on the device so that the loop never synchronises the host)
 * ---------------------------------------------------------------------------------- */

/* W += P @ alpha ; R -= AP @ alpha            (rlaopt/solvers/pcg.py:64-70) */
int rk_pcg_update_wr(int dtype, int64_t n, int64_t k, void* W, int64_t ldw, void* R, int64_t ldr,
                     const void* P, int64_t ldp, const void* AP, int64_t ldap,
                     const double* alpha, void* stream);
/* P = Z + P @ beta                            (rlaopt/solvers/pcg.py:82-83) */
int rk_pcg_update_p(int dtype, int64_t n, int64_t k, void* P, int64_t ldp,
                    const void* Z, int64_t ldz, const double* beta, void* stream);
/* G (k1 x k2, FLOAT64, row-major, device) = X^T Y for tall X (n x k1), Y (n x k2), k1, k2 <= 4,
 * float64 accumulation, fixed-order two-stage reduction: the block inner products P^T AP,
 * R^T Z (rlaopt/solvers/pcg.py:61,79) and the column norms of the residual metric
 * (rlaopt/models/linsys.py:96-99) in two tiny launches. */
size_t rk_gram_small_ws_bytes(int64_t n, int64_t k1, int64_t k2);
int rk_gram_small(int dtype, const void* X, int64_t ldx, const void* Y, int64_t ldy, int64_t n,
                  int64_t k1, int64_t k2, double* G, void* ws, size_t ws_bytes, void* stream);
/* Y[:, c] = a[c] * X[:, c] + b[c] * Y[:, c]  (n x k, any k >= 1) with per-column
 * coefficients in device FLOAT64 arrays (a == NULL means 1, b == NULL means 0: Y is then
 * write-only).  The column-scaled updates of the LSQR recurrences, kept on the device so
 * the iteration never reads a scalar back (no reference counterpart: SURVEY 8a row a14). */
int rk_axpby_cols(int dtype, int64_t n, int64_t k, const void* X, int64_t ldx, const double* a,
                  void* Y, int64_t ldy, const double* b, void* stream);
/* The n-space half of an LSQR step (Paige & Saunders 1982, section 4, steps 3(b), 4, 5) in one
 * launch, per column c of a block solve (G, V, W, Y: n x k row-major; scalars: float64 k-vectors on
 * the device, updated in place):
 *     v <- g - beta v;  alpha = |v|_2;  v <- v / alpha
 *     rho = hypot(rhobar, beta);  cs = rhobar / rho;  sn = beta / rho;  theta = sn alpha
 *     rhobar <- -cs alpha;  phi = cs phibar;  phibar <- sn phibar
 *     y <- y + (phi / rho) w;  w <- v - (theta / rho) w
 * No reference counterpart; replaces ~20 element-wise launches per iteration. */
int rk_lsqr_tail(int dtype, int64_t n, int64_t k, const void* G, int64_t ldg, void* V, int64_t ldv,
                 void* W, int64_t ldw, void* Y, int64_t ldy, const double* beta, double* rhobar,
                 double* phibar, double* alpha, void* stream);
/* One pass over a tall A (m x n, row-major) for BOTH products of a Golub-Kahan / LSQR step:
 *     U  <- A Z - U diag(coef)      (m x k, in place; coef: k float64 scalars on the device)
 *     G   = A^T U_new               (n x k, float64)
 *     ss  = column sums of U_new^2  (k, float64)
 * A^T u for the normalised u = U_new / sqrt(ss) is G / sqrt(ss): the norm is not needed inside
 * the pass, so A is read once per iteration instead of twice (rk_matmat_skinny followed by
 * rk_matmat_skinny_t).  Supported when `rk_normal_pass_supported` says so: k <= 2, n and lda
 * multiples of the 16-byte vector width, n * sizeof <= 4 KiB; otherwise use the two products.
 * No reference counterpart (the reference has no LSQR: SURVEY 8a row a14); the arithmetic
 * follows Paige & Saunders 1982, section 4, steps 3(a)-(b). */
int rk_normal_pass_supported(int dtype, int64_t m, int64_t n, int64_t lda, int64_t k);
size_t rk_normal_pass_ws_bytes(int dtype, int64_t m, int64_t n, int64_t k);
int rk_normal_pass(int dtype, const void* A, int64_t m, int64_t n, int64_t lda, const void* Z,
                   int64_t k, int64_t ldz, void* U, int64_t ldu, const double* coef, double* G,
                   double* ss, void* ws, size_t ws_bytes, void* stream);
/* X = A^-1 B for k x k A and k x nrhs B (FLOAT64, device), LU with partial pivoting,
 * k <= 16, nrhs <= 16 (torch.linalg.solve at rlaopt/solvers/pcg.py:61,80). A and B are
 * not modified; X may alias B. */
int rk_small_solve(int64_t k, int64_t nrhs, const double* A, const double* B, double* X,
                   void* stream);

/* ------------------------------------------------------------------------------------
 * Sparse-sign sketch  [a7, a10]
 * ---------------------------------------------------------------------------------- */

/* Y (s x d) = scale * Omega @ A for the sparse-sign embedding Omega (s x m, <= zeta signed
 * entries per column; rlaopt/sketches/sparse.py:52-76) given in row-sorted form: entries
 * row_ptr[r] .. row_ptr[r+1]-1 of `col` (int32 row index into A) and `sgn` (+1 / -1) belong to
 * sketch row r.  Replaces the dense `Omega_mat @ A` of rlaopt/sketches/sketch.py:75-84 in the
 * SkPre update (rlaopt/preconditioners/skpre.py:84): zeta passes over A instead of an
 * s x m x d GEMM, and no dense s x m matrix. */
int rk_sparse_sign_apply(int dtype, const void* A, int64_t m, int64_t d, int64_t lda,
                         const int* col, const signed char* sgn, const int64_t* row_ptr, int64_t s,
                         double scale, void* Y, int64_t ldy, void* stream);

/* ------------------------------------------------------------------------------------
 * Gaussian test matrices  [a7]
 * ---------------------------------------------------------------------------------- */

/* out (rows x cols, leading dim ld) = scale * N(0,1), Philox4x32-10 counter RNG +
 * Box-Muller, reproducible for (seed, offset) independent of launch geometry.  The
 * device-side replacement for `torch.randn(s, d) / sqrt(s)` in
 * rlaopt/sketches/gauss.py:46-48. */
int rk_randn(int dtype, void* out, int64_t rows, int64_t cols, int64_t ld,
             uint64_t seed, uint64_t offset, double scale, void* stream);

/* out[0 .. k) = k DISTINCT integers drawn uniformly from [0, n): the first k images of a keyed
 * pseudo-random permutation of [0, n) (Feistel network on ceil(log2 n) bits with cycle walking),
 * reproducible for (n, seed) and independent of k and of the launch geometry (a longer draw
 * extends a shorter one).  One launch; replaces the block draw
 * `torch.multinomial(probs, blk_sz, replacement=False)` with uniform probs of rlaopt/solvers/sap.py:76-86
 * (a uniformly random subset; the reference's generator stream is not reproduced). */
int rk_sample_distinct(int64_t n, int64_t k, uint64_t seed, int64_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RK_H_ */
