/*
 * svgpvae_hip.h - C ABI of libsvgpvae_hip.so, the MI355X (gfx950) implementation of the
 * SVGPVAE_Hensman / SVGPVAE_Titsias training step of ratschlab/SVGP-VAE (rotated-MNIST path), of the
 * building blocks of its SPRITES path, and of the N-sized float32 statistics pass of its test pipeline.
 *
 * The reference has no native/FFI layer (it is TensorFlow-1.15 graph code), so each entry point
 * below cites the reference Python it replaces (file:line into the reference checkout) instead of
 * a reference FFI declaration.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = svgp_status; svgp_last_error() gives a
 *     thread-local message.  No C++ exception crosses the boundary.
 *   - all tensor pointers are CALLER-OWNED DEVICE pointers (row-major, contiguous; float64 everywhere
 *     except the svgp_stream_*_f32 entry points); nothing is retained after return.  Scratch lives in the caller-provided workspace whose
 *     layout svgp_mnist_ws_layout() describes (offsets in float64 elements).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it,
 *     no hidden synchronisation, no allocation -> every call is hipGraph-capturable.
 *   - per-step scalars (GECO state, Adam step, lr, beta, alpha) live in a small DEVICE state
 *     vector (svgp_state_slot) so that a captured graph can be replayed unchanged.
 *   - re-entrant, usable from several threads on different streams.  Process-wide state is limited to
 *     what is created once and then only read -- the dlopen'd RCCL entry points (svgp_comm_*) -- and one pair of
 *     library-owned side streams + events per (device, caller stream), created at the first m > 64 step (or with
 *     SVGP_SIDE_STREAMS=1) on that stream and used only by calls that pass that stream.
 */
#ifndef SVGPVAE_HIP_H
#define SVGPVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVGP_VERSION_MAJOR 0
#define SVGP_VERSION_MINOR 1

typedef enum {
    SVGP_OK = 0,
    SVGP_ERR_INVALID = -1,     /* bad argument / null pointer / inconsistent shapes            */
    SVGP_ERR_UNSUPPORTED = -2, /* shape outside what this build implements (e.g. m > 2048)     */
    SVGP_ERR_HIP = -3,         /* a HIP runtime call failed (message has hipGetErrorString)    */
    SVGP_ERR_COMM = -4         /* an RCCL call failed (message has ncclGetErrorString)         */
} svgp_status;

/* Shapes + constants of one rank's share of a rotated-MNIST SVGPVAE step.
 * Mirrors the constructor arguments of mnistSVGP (SVGPVAE_model.py:383-384), mnistVAE
 * (VAE_utils.py:103) and the flags consumed by forward_pass_SVGPVAE (SVGPVAE_model.py:823-825). */
typedef struct {
    int32_t b;             /* rows of THIS rank's batch                                         */
    int32_t b_global;      /* rows of the global batch (= b on one GPU); N_train/b_global is c  */
    int32_t m;             /* number of inducing points (rows of inducing_index_points)         */
    int32_t L;             /* latent channels (mnistVAE L)                                      */
    int32_t M;             /* GPLVM / object-vector dimension (--M)                             */
    int32_t n_obj;         /* rows of the object_vectors table; 0 = no table (aux cols 2: used) */
    int32_t normalize_obj; /* K_obj_normalize (SVGPVAE_model.py:465-474)                        */
    int32_t clip_qs;       /* clipping_qs: clip qnet_var to [1e-3, 10] (SVGPVAE_model.py:858)   */
    int32_t geco;          /* GECO objective (SVGPVAE_model.py:908-915) else beta-ELBO (:917)   */
    int32_t train_ip;      /* 1: inducing points trainable (--ip_joint)                         */
    int32_t train_gp;      /* 1: l_GP, amplitude trainable (--GP_joint)                         */
    int32_t train_ov;      /* 1: object_vectors trainable (--ov_joint)                          */
    int32_t b_cap;         /* row capacity the workspace LAYOUT is computed for (0 = b); keeps    */
                           /* offsets fixed across ragged batches / train-vs-test row counts      */
    int32_t clip_pv;       /* 1: clip the posterior variance p_v to [1e-4, 100] (SPRITES,          */
                           /* SVGPVAE_model.py:891-892); the clip mask enters the reverse pass.    */
                           /* 2: moving ball (:693): only the sampling step uses clip(p_v, 1e-4,   */
                           /* 1000); p_v itself and the cross-entropy keep the raw value           */
    int32_t n_pix;         /* pixels per image in the reconstruction loss (0 = 784 = MNIST)        */
    int32_t titsias;       /* 1: mainSVGP(titsias=True): inside-ELBO = sum_l L_2 (SVGPVAE_model.py:246-259,  */
                           /* 882-883) instead of the Hensman L_3 / KL pair                          */
    int32_t kl_form;       /* 0: Hensman KL of mainSVGP (:270-279).  1: KL of the moving-ball SVGP (:128-137), which  */
                           /* has A_hat where mu_hat is meant and reduces over the whole batch of videos (= the L      */
                           /* channels here): KL_l = 1/2 [.. + L tr(K_mm^-1 A_l A_l)], same sum over l as the          */
                           /* reference; the KL field is then [KL (L) | tr(K_mm^-1 A_l A_l) (L)]                       */
    int32_t single_stat_block; /* m <= 64: 1 = the statistics launches write ONE block per channel instead of 4 row      */
                           /* partials (optional under data parallelism: by default the partial blocks are exchanged  */
                           /* as one message).  m > 64: 1 = the batch is sharded over ranks: the workspace carries    */
                           /* the wire buffer of the packed exchange (xpack)                                          */
    int32_t gemm_f32;      /* large-m path (m > 64) only.  1: every product of the GP block runs on the float32 MFMA  */
                           /* (svgp_dgemm_f32c_batched: float64 storage, float32 products and sums) -- the arithmetic */
                           /* of the reference's float32 SPRITES graph (SVGPVAE_model.py:516); 2: only the statistics */
                           /* S_l = K_mn diag(w) K_nm (sums of non-negative terms for the forward weights) do.  The   */
                           /* m x m factorisations / inverses stay float64 in every mode               */
    int32_t split_grad_exchange; /* data parallelism, row-sharded schedule: 1 = the closing gradient all-reduce in TWO parts   */
                           /* -- decoder + GP parameters + scalar sums as soon as the kernel-matrix reverse pass is done (they   */
                           /* do not depend on the encoder's reverse pass; svgp_mnist_train_step_dp issues it on a side        */
                           /* stream), the encoder's part behind its reverse pass -- so that one of the three small-message    */
                           /* latencies hides under the encoder's reverse pass.  0 (default): one all-reduce.  (The field sits */
                           /* in the padding in front of N_train: the struct's size and every other offset are unchanged.)    */
    double  N_train;       /* mainSVGP.N_train                                                  */
    double  jitter;        /* mainSVGP.jitter                                                   */
    double  kappa_squared; /* GECO kappa^2                                                      */
    double  alpha;         /* GECO moving-average alpha used from the 2nd step on               */
    double  rep_weight;    /* weight of rank-replicated gradient terms: 1 on rank 0, else 0     */
} svgp_mnist_cfg;

/* Flat parameter vector theta (float64).  Offsets in elements.  Order = the reference's
 * trainable-variable creation order: VAE encoder, decoder (VAE_utils.py:114-141), inducing points
 * (SVGPVAE_model.py:203), l_GP, amplitude (:412-413), object_vectors (:421).  TF layouts:
 * conv kernels (kh,kw,cin,cout), dense (in,out), NHWC flatten.                                  */
typedef struct {
    int64_t enc_c1_w, enc_c1_b, enc_c2_w, enc_c2_b, enc_c3_w, enc_c3_b, enc_d_w, enc_d_b;
    int64_t dec_d_w, dec_d_b, dec_c1_w, dec_c1_b, dec_c2_w, dec_c2_b, dec_c3_w, dec_c3_b;
    int64_t ip, l_GP, amplitude, ov;
    int64_t n_enc;   /* number of encoder parameters (prefix of theta)       */
    int64_t n_vae;   /* encoder + decoder                                     */
    int64_t n_total; /* everything                                            */
} svgp_mnist_param_layout;

/* Workspace (float64) offsets of every intermediate of the step.  All are inspectable for tests. */
typedef struct {
    /* encoder (VAE_utils.py:114-126,143-152) */
    int64_t enc_a1, enc_a2, enc_a3;       /* post-ELU activations (b,13,13,8) (b,6,6,8) (b,32)    */
    int64_t qnet_mu, qnet_var_raw, qnet_var; /* (b,L) each; var = clip(exp(.))                    */
    /* kernel matrices (SVGPVAE_model.py:427-476) */
    int64_t K, Kn, knn;                   /* (m,m) (b,m) (b); m > 64: W = Kn Ki K (b,m) behind the b rows of Kn */
    /* forward statistics: ONE contiguous all-reduce block [S | v] (titsias: [S | v | tit_S2 | tit_v2]).
     * S (P,L,m,m), v (P,L,m) with P = stat_parts ROW PARTIALS (the LDS-resident path splits the rows of every channel
     * over P workgroups; the statistics are the sums over p, taken by the consumers on load; 1 for m > 64 and
     * with cfg.single_stat_block, which the engines set when the blocks travel through an all-reduce).  The
     * backward block [A2 | ud | td] has the same partial structure.  Sums over ranks commute with the sum over p. */
    int64_t statA, statA_len, S, v, stat_parts;
    /* m x m factor stage (SVGPVAE_model.py:239,270-279,319-341) */
    int64_t Ki, ldK, Si, t, G, A, Aji, mu_hat, u, M2, KL, q; /* M2 = Ki A Ki (L,m,m; m <= 64 only); q (b) */
    /* per-sample stage (:264-299,332-337, 888-902; utils.py:498-504) */
    int64_t p_m, p_v, e, d, eps, z;       /* (b,L) each                                           */
    /* decoder (VAE_utils.py:128-141,154-162) */
    int64_t dec_h0, dec_a1, dec_a2, recon; /* (b,128) (b,8,8,8) (b,14,14,8) (b,28,28,1)           */
    /* pre-activation gradients of the decoder layers, written by svgp_mnist_decoder_bwd_data for the weight half */
    int64_t dec_d2, dec_d1, dec_dh0;      /* (b,14,14,8) (b,8,8,8) (b,128)                        */
    int64_t flags;                        /* (64) u64 counters of the intra-launch hand-offs: [0..2] svgp_mnist_encoder_bwd_km_sum (produced,
                                           * consumed, sticky error), [8, 8+L) svgp_gp_stats_factor_bwd_wgrad (one per channel); the
                                           * workspace must be ZERO here before its first use (every call leaves them zero) */
    int64_t dec_weff;                     /* (2176) effective parity-class weights of the decoder's three up-convolutions for the
                                           * current theta: written by svgp_mnist_encoder_kernel_matrix_fwd, read by the `_pre` forms */
    /* backward */
    int64_t zbar, g_pv, g_pm, mvbar;      /* (b,L) each                                           */
    int64_t statB, statB_len, A2, ud, td; /* ONE contiguous all-reduce block [A2 | ud | td]       */
    int64_t Kbar, fb_part, Qm, vbar, Ssym; /* (m,m) (2,L,m,m) scratch (L,m,m) (L,m) (L,m,m)        */
    int64_t Knbar_part;                   /* (L,b,m) per-channel row gradients before the sum     */
    int64_t scr_bm, scr_mm, scr_vec, scr_inv, scr_bl; /* scratch of the large-m (m > 64) path; scr_bm also holds the
                                           * forward products [Kn; W] Si_l, Kn Ki that the reverse pass re-reads */
    int64_t scr_sm;                       /* m > 64: 10 channel-independent (m,m) matrices (K Ki, Kn^T Wbar, ..) */
    int64_t Knbar, knnbar, ybar, s2bar;   /* (b,m) (b) (b,L) (b,L)                                */
    int64_t d_on;                         /* (b,M) gradient of gathered object rows               */
    /* partial sums */
    int64_t part_dec, part_enc, n_part;   /* (n_part, n_dec) / (n_part, n_enc) weight-grad partials */
    int64_t part_gp;                      /* (m + n_postblk, 2) amplitude / length-scale partials */
    int64_t part_sums, n_post;            /* (n_part,4) [.,.,recon sq,.] then (n_post,2) [L3 data, CE] */
    /* final all-reduce block: [grad (n_total) | sums (8)] */
    int64_t gradC, gradC_len, grad, sums;
    /* Titsias branch (cfg.titsias; zero-sized otherwise): statistics with weights 1/(var + jitter) inside statA,
     * (K + jI + S2)^-1, its product with v2, scalars [logdet (L) | v2.t2 (L) | row sum (1)] */
    int64_t tit_S2, tit_v2, tit_Si, tit_t, tit_scal;
    /* m > 64 with cfg.single_stat_block (batch sharded over ranks): one tile-packed buffer of L symmetric matrices
     * (svgp_sym_pack): what the channel-sharded exchange moves instead of a full (L,m,m) block.
     * xpack_len = L * svgp_sym_packed_elems(m) (0 otherwise).                                                          */
    int64_t xpack, xpack_len;
    int64_t total;                        /* workspace size in float64 elements                   */
} svgp_mnist_ws_layout;

/* Device state vector (float64[SVGP_STATE_LEN]); the host initialises slots 0..5, kernels keep
 * them up to date (MNIST_experiment.py:313-355 host state machine moved on device).            */
typedef enum {
    SVGP_ST_C_MA = 0,        /* GECO moving average C_ma  (init 0.0, MNIST_experiment.py:314)    */
    SVGP_ST_LAGRANGE = 1,    /* GECO lagrange multiplier  (init 1.0, :315)                       */
    SVGP_ST_ALPHA = 2,       /* alpha used by the NEXT step (init 0.0 = first_step, :330-333)    */
    SVGP_ST_ADAM_T = 3,      /* number of Adam updates done so far (global_step)                 */
    SVGP_ST_LR = 4,          /* learning rate                                                    */
    SVGP_ST_BETA = 5,        /* beta of the beta-ELBO                                            */
    SVGP_ST_ELBO = 6,        /* outputs of the last step (the scalar members of the 16-tuple)    */
    SVGP_ST_RECON_LOSS = 7,
    SVGP_ST_KL_TERM = 8,
    SVGP_ST_INSIDE_ELBO = 9,
    SVGP_ST_CE_TERM = 10,
    SVGP_ST_INSIDE_RECON = 11,
    SVGP_ST_INSIDE_KL = 12,
    SVGP_ST_RNG_CTR = 13,    /* counter of the on-device N(0,1) generator (bit pattern of u64)   */
    SVGP_ST_TICKET = 15,     /* u64 workgroup ticket of svgp_adam_tf1_finalize; 0 between launches */
    SVGP_STATE_LEN = 16
} svgp_state_slot;

int         svgp_version(void);
const char* svgp_last_error(void);
/* sizeof() of the ABI structs as compiled into the library, so a binding can verify its own mirror of a struct at load
 * time (a stale mirror shifts every later field silently): which = 0 svgp_mnist_cfg, 1 svgp_mnist_param_layout,
 * 2 svgp_mnist_ws_layout, 3 svgp_stream_kdesc, 4 svgp_conv_desc, 5 svgp_sprites_kcfg, 6 svgp_pearce_bufs; -1 otherwise. */
int         svgp_struct_sizeof(int which);

int svgp_mnist_param_layout_get(const svgp_mnist_cfg* cfg, svgp_mnist_param_layout* out);
int svgp_mnist_ws_layout_get(const svgp_mnist_cfg* cfg, svgp_mnist_ws_layout* out);

/* ---- stage entry points (each enqueues 1-2 kernels on `stream`) ----------------------------
 * theta: flat parameters; ws: workspace; images (b,28,28,1); aux (b,2+M); state: device state.   */

/* mnistVAE.encode + clip  (VAE_utils.py:143-152, SVGPVAE_model.py:854-859) */
int svgp_mnist_encoder_fwd(const svgp_mnist_cfg*, const double* theta, const double* images,
                           double* ws, void* stream);
/* mnistSVGP.kernel_matrix x3: K_mm, K_nm, diag K_nn (SVGPVAE_model.py:427-476) */
int svgp_kernel_matrix_fwd(const svgp_mnist_cfg*, const double* theta, const double* aux,
                           double* ws, void* stream);
/* mnistSVGP.kernel_matrix(x, y, x_inducing, y_inducing, diag_only) for ARBITRARY row sets (SVGPVAE_model.py:427-476):
 * x (nx, 2+M), y (ny, 2+M), rows [id, angle, o_1..o_M].  x_gather / y_gather != 0: that side's object vector is
 * table[(int)row[0]] (the reference's `not *_inducing` with a GPLVM table, :451,455), else the row's columns 2:.
 * diag_only: out (nx) = k(x_i, y_i), nx == ny (kernel.apply, :458-467); else out (nx, ny) row-major.                 */
int svgp_kernel_matrix_xy(int M, int normalize, int nx, const double* x, int x_gather, int ny, const double* y,
                          int y_gather, const double* table, const double* l_GP, const double* amplitude,
                          int diag_only, double* out, void* stream);
/* S_l = K_mn diag(1/var_l) K_nm, v_l = K_mn (y_l/var_l)  (SVGPVAE_model.py:328-334) and, for m <= 64 in the
 * same launch, K_mm_inv / logdet (:239,270,273) (m > 64: svgp_gp_factor_fwd forms them next to the channel
 * inverses).  Output block statA is what DP all-reduces.                                           */
int svgp_gp_stats_fwd(const svgp_mnist_cfg*, double* ws, void* stream);
/* Sigma_l^-1, mu_hat, A_hat, KL_l (SVGPVAE_model.py:331-341, 270-279) */
int svgp_gp_factor_fwd(const svgp_mnist_cfg*, double* ws, void* stream);
/* posterior mean/var at the batch points, L3 / cross-entropy integrands, z = p_m + eps sqrt(p_v)
 * (SVGPVAE_model.py:264-265,284-299,332-337,895-902).  eps (b,L) may be NULL: then N(0,1) numbers
 * are drawn on device (Philox) as tf.random.normal does at :901.                                  */
int svgp_gp_posterior_fwd(const svgp_mnist_cfg*, const double* eps, double* ws, double* state,
                          void* stream);
/* mnistVAE.decode + squared reconstruction error (VAE_utils.py:154-162, SVGPVAE_model.py:905-918) */
int svgp_mnist_decoder_fwd(const svgp_mnist_cfg*, const double* theta, const double* images,
                           double* ws, void* stream);
/* reverse of the decoder; writes zbar and decoder weight-gradient partials */
int svgp_mnist_decoder_bwd(const svgp_mnist_cfg*, const double* theta, const double* images,
                           double* ws, const double* state, void* stream);
/* The reverse pass of the decoder in two halves (tf.gradients of VAE_utils.py:128-141,154-162; MNIST_experiment.py:202-205).
 * _data: the chain d recon -> zbar that the GP reverse stages wait for; also stores the pre-activation gradients ws.dec_d2 /
 * dec_d1 / dec_dh0.  _weights: the decoder weight-gradient partials (ws.part_dec) from those and the stored activations; feeds
 * only svgp_mnist_grad_reduce, so it may run anywhere between _data and the reduction -- svgp_gp_factor_bwd_nofinal_wgrad runs
 * it as extra workgroups of the reverse factor launch (m <= 64), which leaves 240 of 256 CUs idle.  threads: 256 or 512 per
 * workgroup (the rider form is 256); n_types: 1, 2 or 3 workgroups per image, each taking a group of layers (the rider form
 * is 3: UpC3 | UpC2 | UpC1 + dense).  _data + _weights == svgp_mnist_decoder_bwd up to summation order. */
int svgp_mnist_decoder_bwd_data(const svgp_mnist_cfg*, const double* theta, const double* images,
                                double* ws, const double* state, void* stream);
int svgp_mnist_decoder_bwd_weights(const svgp_mnist_cfg*, const double* images, double* ws, const double* state,
                                   int threads, int n_types, void* stream);
/* svgp_mnist_decoder_fwd / _bwd_data that LOAD the effective up-convolution weights (UpSampling2D + 3x3 as four parity-specific
 * 2x2 convolutions) from ws.dec_weff instead of rebuilding them in each of their workgroups.  Valid only behind
 * svgp_mnist_encoder_kernel_matrix_fwd with the SAME theta on the same workspace (svgp_mnist_step_phase issues them so). */
int svgp_mnist_decoder_fwd_pre(const svgp_mnist_cfg*, const double* theta, const double* images, double* ws, void* stream);
int svgp_mnist_decoder_bwd_data_pre(const svgp_mnist_cfg*, const double* theta, const double* images,
                                    double* ws, const double* state, void* stream);
/* Titsias branch (cfg.titsias = 1), SVGPVAE_model.py:246-259: L_2 = -1/2 [b log 2pi + log det C + y^T C^-1 y +
 * sum_n (k_nn - q_n)/var_n], C = diag(var) + K_nm (K_mm + jI)^-1 K_mn + jI (b x b).  Computed in m x m space through
 * the Woodbury identity with the statistics S2_l = sum_n k_n k_n^T/(var_nl + j), v2_l = sum_n y_nl k_n/(var_nl + j):
 *   log det C = sum_n log(var_n + j) + log det(K + jI + S2) - log det(K + jI)
 *   y^T C^-1 y = sum_n y_n^2/(var_n + j) - v2^T (K + jI + S2)^-1 v2
 * so the b x b inverse and Cholesky of the reference never exist and the branch shards over rows like the Hensman
 * one (S2, v2 ride in the statA all-reduce).  p_m, p_v, the cross-entropy term and everything downstream are the
 * shared stages; with cfg.titsias the L_3 / KL seeds of their reverse passes are zero and
 * svgp_gp_titsias_bwd adds the L_2 gradients to ybar, s2bar, Knbar, knnbar, Kbar.
 *   svgp_gp_titsias_stats : after svgp_gp_stats_fwd (phase 0)        -> ws[tit_S2], ws[tit_v2]
 *   svgp_gp_titsias_fwd   : after svgp_gp_posterior_fwd (phase 1)    -> ws[tit_Si], ws[tit_t], ws[tit_scal]
 *   svgp_gp_titsias_bwd   : after svgp_gp_posterior_bwd (phase 2), before svgp_kernel_matrix_bwd            */
int svgp_gp_titsias_stats(const svgp_mnist_cfg*, double* ws, void* stream);
int svgp_gp_titsias_fwd(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
int svgp_gp_titsias_bwd(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* upstream gradients of p_m, p_v + backward statistics block statB (DP all-reduces it) */
int svgp_gp_stats_bwd(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* all m x m reverse algebra -> Kbar and the per-channel matrices of the row stage */
int svgp_gp_factor_bwd(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* row-local gradients: Knbar, knnbar, ybar, s2bar */
int svgp_gp_posterior_bwd(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* VJP of kernel_matrix: gradients of inducing points, l_GP, amplitude, object_vectors into grad */
int svgp_kernel_matrix_bwd(const svgp_mnist_cfg*, const double* theta, const double* aux,
                           double* ws, void* stream);
/* reverse of the encoder (through exp and the clip mask); encoder weight-gradient partials */
int svgp_mnist_encoder_bwd(const svgp_mnist_cfg*, const double* theta, const double* images,
                           double* ws, void* stream);
/* fixed-order reduction of all partials into the final block [grad | sums] (DP all-reduces it) */
int svgp_mnist_grad_reduce(const svgp_mnist_cfg*, double* ws, void* stream);
/* What the training phases use instead of the pair (svgp_kernel_matrix_bwd, svgp_mnist_grad_reduce): the
 * kernel-matrix VJP without its last launch (the deterministic scatter of the gathered-row gradients into the object
 * table and the two hyper-parameter sums), and the reduction launch with those workgroups appended -- one dependent
 * launch less on the critical path.  Results are identical. */
int svgp_kernel_matrix_bwd_partials(const svgp_mnist_cfg*, const double* theta, const double* aux, double* ws,
                                    void* stream);
int svgp_mnist_grad_reduce_all(const svgp_mnist_cfg*, const double* aux, double* ws, void* stream);
/* part 1: everything of svgp_mnist_grad_reduce_all except the encoder's weights (decoder weights, scalar sums, object-table
 * scatter and GP hyper-parameter sums) -> gradC[n_enc:]; part 2: the encoder's weights -> gradC[:n_enc].  1 + 2 == _all.
 * svgp_mnist_step_phase phases 4 / 5 = phase 2 cut at that point (4: ... kernel-matrix reverse pass, part 1; 5: encoder reverse
 * pass, part 2) for cfg.split_grad_exchange. */
int svgp_mnist_grad_reduce_part(const svgp_mnist_cfg*, const double* aux, double* ws, int part, void* stream);
/* Two more phase-form pairs with identical results (m <= 64; for larger m they are the plain stages).  A piece of work
 * that is off the critical path moves into extra workgroups of a later launch that leaves most CUs idle:
 *   (A_hat + jI)^-1 and the log det term of KL:  svgp_gp_factor_fwd_defer_aji ... svgp_gp_stats_bwd_with_aji
 *   the channel sum Kbar:                        svgp_gp_factor_bwd_nofinal   ... svgp_gp_posterior_bwd_with_final */
int svgp_gp_factor_fwd_defer_aji(const svgp_mnist_cfg*, double* ws, void* stream);
/* svgp_mnist_encoder_fwd + svgp_kernel_matrix_fwd (independent of each other) in one launch: the kernel-matrix
 * elements are computed by extra workgroups of the encoder launch */
int svgp_mnist_encoder_kernel_matrix_fwd(const svgp_mnist_cfg*, const double* theta, const double* images,
                                         const double* aux, double* ws, void* stream);
int svgp_gp_stats_bwd_with_aji(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* what the training phases use since round 2 for the same deferred work: svgp_gp_factor_fwd_defer_aji ...
 * svgp_gp_posterior_fwd_with_aji (L more workgroups of the row-stage launch, which has several workgroups per CU in
 * flight) ... svgp_gp_stats_bwd.  Results identical. */
int svgp_gp_posterior_fwd_with_aji(const svgp_mnist_cfg*, const double* eps, double* ws, double* state, void* stream);
/* m > 64: svgp_gp_factor_fwd_defer_aji leaves out the whole tail of the stage -- (A_hat_l + jI)^-1, its log det and the KL_l
 * scalars -- and this call performs it.  It may be issued on a DIFFERENT stream (ordered after svgp_gp_factor_fwd_defer_aji,
 * joined before svgp_gp_factor_bwd / the final scalars): it shares no buffer with svgp_gp_posterior_fwd, the decoder and
 * svgp_gp_stats_bwd, so the second batched inverse of the step runs beside them (svgp_mnist_step_phase does this on a
 * library-owned side stream unless SVGP_SIDE_STREAMS=0). */
int svgp_gp_factor_fwd_aji_tail(const svgp_mnist_cfg*, double* ws, void* stream);
/* m > 64: svgp_gp_factor_bwd in two halves.  _early (T1 = S Ki, Ki S Ki, T1 A_hat, Abar, Gbar = K Abar, Z = Sigma^-1 Gbar,
 * Gbar K: 5.5 of the stage's 8 m^3 L products) needs forward quantities, the loss seeds of `state` and (A_hat + jI)^-1 only --
 * no reverse statistic -- and may run on the stream of svgp_gp_factor_fwd_aji_tail, behind it; _late needs _early, A2 / ud /
 * td (svgp_gp_stats_bwd and their exchange).  _early + _late == svgp_gp_factor_bwd, operation for operation. */
int svgp_gp_factor_bwd_early(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
int svgp_gp_factor_bwd_late(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* _early = _early_a + _early_b: _early_a (T1, Ki S Ki, T1 A_hat) does not even need (A_hat + jI)^-1 and may run beside
 * svgp_gp_factor_fwd_aji_tail on a third stream; _early_b (Abar, Gbar, Z, Gbar K) is ordered behind both. */
int svgp_gp_factor_bwd_early_a(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
int svgp_gp_factor_bwd_early_b(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* _late = _late_a + _late_b: _late_a (the vector chain; with the statistic SW formed by svgp_gp_stats_bwd -- all rows local,
 * b < 3 m -- also X, vbar, Si X and -(Si X) Si) reads nothing the early half writes, so a caller that runs the early half on
 * another stream issues it BEFORE joining that stream; _late_b follows the join.
 * WRITE SET of _late_a: the scratch vectors ubar / mubar / tbar always; in the row form (cfg.b == cfg.b_global,
 * cfg.b < 3 m, or cfg.titsias) also ws.vbar, the Ssym slot (it parks Si X there) and scratch matrix 1 (X, then -(Si X) Si) --
 * none of which the early half touches IN THAT FORM (it then skips its own T = S P in scratch matrix 1).  The form is a function
 * of cfg.b / cfg.b_global / cfg.titsias alone and is evaluated by svgp_gp_stats_bwd, _early and _late_a independently: a caller
 * must not change those fields between the three calls of one step (svgp_mnist_step_phase / _train_step never do; row-form X0
 * itself is left in scratch matrix 2 by svgp_gp_stats_bwd, so the stage may be repeated on a workspace). */
int svgp_gp_factor_bwd_late_a(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
int svgp_gp_factor_bwd_late_b(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* _late_b in three pieces (round 6): _channels (the channel block: X sandwiches unless _late_a took them, Ssym, the channel sum) and _kbar
 * (the single-matrix chain of the gradient of Ki: K Pbar^T, Kib, Ki Kib Ki, Pbar^T Ki -- five small launches that read nothing of
 * _channels) may run on two streams; _final assembles Kbar and follows both.  _late_b == _channels; _kbar; _final. */
int svgp_gp_factor_bwd_late_b_channels(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
int svgp_gp_factor_bwd_late_b_kbar(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
int svgp_gp_factor_bwd_late_b_final(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
int svgp_gp_factor_bwd_nofinal(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* m <= 64: svgp_kernel_matrix_bwd_partials + svgp_mnist_encoder_bwd in ONE launch (the two are independent: both consume the
 * reverse row stage; tf.gradients of SVGPVAE_model.py:427-476 and VAE_utils.py:112-126,143-152).  Results identical. */
int svgp_mnist_encoder_bwd_km(const svgp_mnist_cfg*, const double* theta, const double* images, const double* aux,
                              double* ws, void* stream);
/* m <= 64, the training step's form: svgp_gp_posterior_bwd = _rows (pass 1: ybar, s2bar, the per-channel partials of Knbar) + pass 2
 * (the sums over channels: Knbar, knnbar, Kbar -- consumed by the kernel-matrix VJP only); _km_sum = pass 2 + the VJP + the encoder's
 * reverse pass in ONE launch: the VJP workgroups wait for the pass-2 workgroups on a counter in ws.flags (release / acquire at
 * agent scope), the image workgroups wait for nobody.  Results identical to the separate launches. */
int svgp_gp_posterior_bwd_rows(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
int svgp_mnist_encoder_bwd_km_sum(const svgp_mnist_cfg*, const double* theta, const double* images, const double* aux,
                                  double* ws, const double* state, void* stream);
/* Diagnostic: registers per lane of the (m 32, M 8) instance of that launch.  A kernel-matrix VJP workgroup and an image workgroup
 * share a CU only while it is <= 168; the kernel carries no occupancy hint (it cost the image code 1.7 us), tests check the number. */
int svgp_mnist_encoder_bwd_km_regs(int* out);
/* m <= 32, training step (round 6): svgp_mnist_decoder_bwd_data_pre + the deferred tail of the forward factor stage -- Aji = (A_hat +
 * jitter I)^-1 and the log det term of KL (SVGPVAE_model.py:271-279), left open by svgp_gp_factor_fwd_defer_aji -- as L rider
 * workgroups at the head of the same launch.  The forward row stage then runs as svgp_gp_posterior_fwd (7.8 us instead of the 10.1 us
 * of svgp_gp_posterior_fwd_with_aji, whose riders bounded it).  Same bits.  _regs: registers per lane of that kernel (<= 168 keeps a
 * rider and an image workgroup on one CU). */
int svgp_mnist_decoder_bwd_data_pre_aji(const svgp_mnist_cfg*, const double* theta, const double* images, double* ws,
                                        const double* state, void* stream);
int svgp_mnist_decoder_bwd_data_aji_regs(int* out);
/* m <= 64: svgp_gp_factor_bwd_nofinal + svgp_mnist_decoder_bwd_weights(threads = 256) in ONE launch: the L channel workgroups
 * first, then min(b, 256) rider workgroups with the decoder's weight-gradient partials (needs svgp_mnist_decoder_bwd_data before). */
int svgp_gp_factor_bwd_nofinal_wgrad(const svgp_mnist_cfg*, const double* images, double* ws, const double* state,
                                     void* stream);
/* ... and svgp_gp_stats_bwd as P L leading workgroups of the same launch (single-GPU step only: nothing may be exchanged between the
 * reverse statistics and the reverse factor stage).  Channel workgroup l waits for its own P producers on ws.flags[8 + l] (write-through
 * payload, drained, relaxed agent-scope counter; the consumer re-arms it).  Same bits as svgp_gp_stats_bwd followed by
 * svgp_gp_factor_bwd_nofinal_wgrad.  L <= 56. */
int svgp_gp_stats_factor_bwd_wgrad(const svgp_mnist_cfg*, const double* images, double* ws, const double* state,
                                   void* stream);
int svgp_gp_posterior_bwd_with_final(const svgp_mnist_cfg*, double* ws, const double* state, void* stream);
/* tf.train.AdamOptimizer.apply_gradients, TF1 formula (MNIST_experiment.py:200,207-208) */
int svgp_adam_tf1_step(int64_t n, double* theta, const double* grad, double* adam_m, double* adam_v,
                       const double* state, double beta1, double beta2, double epsilon, void* stream);
/* scalar members of the 16-tuple + GECO state update + global_step (SVGPVAE_model.py:880-925,
 * MNIST_experiment.py:334-355) */
int svgp_elbo_finalize(const svgp_mnist_cfg*, double* ws, double* state, void* stream);
/* svgp_adam_tf1_step + svgp_elbo_finalize in one launch (what phase 3 uses) */
int svgp_adam_tf1_finalize(const svgp_mnist_cfg*, int64_t n, double* theta, const double* grad, double* adam_m,
                           double* adam_v, double* ws, double* state, double beta1, double beta2, double epsilon,
                           void* stream);
/* same, but global_step is not advanced (forward/backward only, no optimiser update) */
int svgp_elbo_finalize_noadam(const svgp_mnist_cfg*, double* ws, double* state, void* stream);

/* ---- phases: the stages above grouped between the data-parallel exchange points -------------
 * phase 0: encoder_kernel_matrix_fwd, gp_stats_fwd                 -> all-reduce ws[statA]
 * phase 1: gp_factor_fwd_defer_aji, gp_posterior_fwd, decoder_fwd, decoder_bwd, gp_stats_bwd_with_aji
 *                                                                   -> all-reduce ws[statB]
 * phase 2: gp_factor_bwd_nofinal, gp_posterior_bwd_with_final, kernel_matrix_bwd_partials, encoder_bwd,
 *          grad_reduce_all
 *                                                                   -> all-reduce ws[gradC]
 * phase 3: adam_tf1_step (skipped when adam_m == NULL), elbo_finalize
 * svgp_mnist_train_step runs phases 0..3 back to back (single GPU).
 * m > 64 without cfg.titsias (unless SVGP_SIDE_STREAMS=0): phase 1 issues svgp_gp_factor_fwd_aji_tail and
 * svgp_gp_factor_bwd_early on a side stream and records that for `ws`; phase 2 joins the branch and runs
 * svgp_gp_factor_bwd_late only when that record exists for the same (stream, ws) -- otherwise (phase-1 stages issued
 * through the individual entry points, switch changed in between, cfg.titsias) it runs the whole reverse factor
 * stage.  With cfg.titsias nothing is forked: svgp_gp_titsias_fwd uses the inverse scratch the tail would use.    */
int svgp_mnist_step_phase(const svgp_mnist_cfg*, int phase, double* theta, const double* images,
                          const double* aux, const double* eps, double* ws, double* state,
                          double* adam_m, double* adam_v, void* stream);
int svgp_mnist_train_step(const svgp_mnist_cfg*, double* theta, const double* images,
                          const double* aux, const double* eps, double* ws, double* state,
                          double* adam_m, double* adam_v, void* stream);

/* ---- data-parallel exchange over RCCL on the compute stream -------------------------
 * The reference is single-process; these are the three sum-exchanges the row-sharded step needs
 * (statistics S_l, v_l of SVGPVAE_model.py:328-334,339-340; their reverse-mode counterparts; the
 * parameter gradients of MNIST_experiment.py:197-208).  One communicator per process (= per GPU).
 * Bootstrap: rank 0 calls svgp_comm_unique_id, the bytes reach the other ranks by any side channel
 * (torch.distributed broadcast_object_list in engine.py), every rank calls svgp_comm_init with its
 * device current.  RCCL itself is resolved with dlopen at the first call (SVGP_ERR_UNSUPPORTED when
 * absent).  svgp_mnist_train_step_dp = phase 0 | all-reduce statA | phase 1 | all-reduce statB |
 * phase 2 | all-reduce gradC | phase 3, all enqueued on `stream` with no host synchronisation;
 * cfg->b is the rank's row count, cfg->b_global the global batch, cfg->rep_weight 1 on one rank.   */
int svgp_comm_unique_id_bytes(void);
int svgp_comm_unique_id(void* out, int nbytes);
int svgp_comm_init(const void* unique_id, int nbytes, int rank, int nranks, void** comm_out);
int svgp_comm_destroy(void* comm);
int svgp_allreduce_sum_f64(void* comm, double* buf, int64_t count, void* stream);
int svgp_allreduce_sum_f32(void* comm, float* buf, int64_t count, void* stream);   /* float32 streaming statistics */
/* Channel-sharded exchange for large (L,m,m) blocks (SURVEY 8e "reduce-scatter over L -> factorize L/G channels per rank
 * -> all-gather"): in-place on a buffer of nranks equal chunks, rank r owning buf[r * count, (r+1) * count).  The
 * schedule (sprites.py / engine.ChannelShardedExchange): reduce-scatter S, v over channels | svgp_gp_factor_fwd_channels
 * on the rank's window | all-gather Sigma_l^-1, M2, t, u, KL | ... | reduce-scatter A2, ud, td |
 * svgp_gp_factor_bwd_channels | all-gather Ssym, vbar.  Kbar needs no exchange: each rank's window share flows
 * through svgp_kernel_matrix_bwd (linear in Kbar) into the gradient all-reduce (cfg.rep_weight = 1 on every rank).    */
int svgp_reduce_scatter_sum_f64(void* comm, double* buf, int64_t count_per_rank, void* stream);
int svgp_allgather_f64(void* comm, double* buf, int64_t count_per_rank, void* stream);
/* the factor stages restricted to the channel window [l0, l0 + nl) (m > 64): (K_mm + jI)^-1 and q_n are formed by every
 * call; with l0 = 0, nl = L they are svgp_gp_factor_fwd / svgp_gp_factor_bwd                                            */
int svgp_gp_factor_fwd_channels(const svgp_mnist_cfg*, int l0, int nl, double* ws, void* stream);
int svgp_gp_factor_bwd_channels(const svgp_mnist_cfg*, int l0, int nl, double* ws, const double* state, void* stream);
/* ... and their parts, the split of the two-stream step on a channel window: forward part 0 = the whole stage, 1 = without
 * the (A_hat + jI)^-1 tail (svgp_gp_factor_fwd_defer_aji), 2 = the tail (svgp_gp_factor_fwd_aji_tail); reverse part 0 = the
 * whole stage, 1 = early half, 2 = late half, 3 / 4 = the two parts of the early half (svgp_gp_factor_bwd_early_a / _b).  */
int svgp_gp_factor_fwd_channels_part(const svgp_mnist_cfg*, int l0, int nl, int part, double* ws, void* stream);
int svgp_gp_factor_bwd_channels_part(const svgp_mnist_cfg*, int l0, int nl, int part, double* ws, const double* state,
                                     void* stream);
/* svgp_mnist_train_step_dp, channel-sharded form (m > 64, L divisible by the rank count): five exchange POINTS, each ONE
 * grouped RCCL launch (ncclGroupStart / End): [S | v] reduce-scatter, [Sigma^-1 | M2 | t | u] all-gather, [A2 | ud | td]
 * reduce-scatter, [Ssym | vbar | KL] all-gather, gradient all-reduce.  The (L,m,m) members are symmetric and travel
 * TILE-PACKED (svgp_sym_pack: lower triangle in 32 x 32 tiles, 52 % of the square at m = 800; SURVEY 8e "halve via
 * symmetry") with SVGP_DP_PACK=1, off with 0; default: from m >= 512.  The tail of the forward factor stage and the early half of the
 * reverse one run on the library's side stream beside the all-gather, the row stage, the networks and the reverse
 * statistics (as in the single-GPU step; SVGP_SIDE_STREAMS=0 puts them in line).                                      */
int svgp_mnist_train_step_dp(const svgp_mnist_cfg*, void* comm, double* theta, const double* images,
                             const double* aux, const double* eps, double* ws, double* state,
                             double* adam_m, double* adam_v, void* stream);
/* Several collectives as one RCCL launch: everything issued on `comm` between _begin and _end (ncclGroupStart / End). */
int svgp_comm_group_begin(void* comm);
int svgp_comm_group_end(void* comm);
/* HIP events around every exchange point of svgp_mnist_train_step_dp (pack + collective(s) + unpack): enable != 0 switches
 * the recording on; svgp_comm_timing_read waits for the last recorded step and returns the microseconds of its points
 * (n <= cap of them, in step order).  bench.py --gpus N reports them as `collectives_us`.                              */
int svgp_comm_timing(void* comm, int enable);
int svgp_comm_timing_read(void* comm, float* us, int cap, int* n);
/* Tile-packed lower triangle of L symmetric m x m matrices: per matrix the nt (nt + 1) / 2 tiles (ti >= tj) of 32 x 32,
 * nt = ceil(m / 32), tile t = ti (ti + 1) / 2 + tj stored row-major (rows / columns >= m are zero).  avg != 0: the tile
 * holds (x_ij + x_ji) / 2 (for products that are symmetric only up to rounding, M2 = Ki A Ki); else x_ij of the lower
 * tiles.  svgp_sym_unpack writes both triangles of the square matrices.  src / dst strides: m * m and packed_elems.   */
int64_t svgp_sym_packed_elems(int m);
int svgp_sym_pack(int m, int L, int avg, const double* src, double* dst, void* stream);
int svgp_sym_unpack(int m, int L, const double* src, double* dst, void* stream);

/* ---- full-data statistics in float32, streamed over N (8f rank 1) -----------------
 * replace precompute_GP_params_SVGPVAE's K_nm build and its per-channel
 *   K_mn (K_nm * 1/var_l)  and  K_mn (mean_l / var_l)        (SVGPVAE_model.py:1004-1017)
 * for N-sized inputs: N = 2^20 rows / m = 2048 per the stress config, or the SPRITES training set
 * (SPRITES_experiment.py:176-182).  float32 is the reference's dtype for SPRITES (tf.float32).
 * kind 0: periodic(angle) x linear(object vector)   mnistSVGP.kernel_matrix   :427-476, p = [l_GP, amplitude]
 * kind 1: linear(action) x linear(character)         spritesSVGP.kernel_matrix :550-600
 * kind 2: SE(action) x SE(character), --K_SE         :530-544,                 p = [l1, sigma1, l2, sigma2]
 * normalize = cosine normalisation of the linear factors (:465-474, :576-598).
 * Batch rows x: kind 0 (n, 2+d2) [id, angle, o..] with o gathered from `table` (n_table, d2) when
 * n_table > 0; kinds 1,2 (n, 1+d2) [action_id, chr..] with the action vector gathered from `table`
 * (n_table, d1).  Inducing rows: kind 0 (m, 2+d2); kinds 1,2 (m, d1+d2).
 * svgp_stream_features_f32 writes feature_elems(n) floats; svgp_stream_knm_f32 writes K_nm (n, m) row-major;
 * svgp_stream_stats_f32 writes S (L, m, m) = K_nm^T diag(1/var_l) K_nm and v (L, m) = K_nm^T (mean_l/var_l),
 * means / vars (n, L) row-major, 1/var via reciprocal_no_nan.  Nothing is added or inverted here.       */
typedef struct {
    int32_t kind, d1, d2, normalize, n_table;
    float   p[4];
} svgp_stream_kdesc;
int64_t svgp_stream_feature_elems(const svgp_stream_kdesc*, int64_t n);
int svgp_stream_features_f32(const svgp_stream_kdesc*, int64_t n, const float* x, int ldx, int inducing,
                             const float* table, float* feat, void* stream);
int svgp_stream_knm_f32(const svgp_stream_kdesc*, int64_t n, int m, const float* feat_rows,
                        const float* feat_inducing, float* K_nm, void* stream);
int64_t svgp_stream_stats_workspace_elems(int64_t n, int m, int L);
int svgp_stream_stats_f32(int64_t n, int m, int L, const float* K_nm, const float* means, const float* vars,
                          float* S, float* v, float* ws, int64_t ws_elems, void* stream);

/* ---- batched float64 linear algebra on device matrices (large-m path; also usable on their own) ---
 * replace tf.matmul / tf.linalg.inv / tf.linalg.cholesky+log(diag) (SVGPVAE_model.py:239,270-274,319,328-341)
 * svgp_dgemm_batched: C[l] = alpha op(A[l]) op(B[l]) + beta C[l]; C is M x N, contraction K; ta/tb = 1 means
 *   the operand is stored transposed (A as K x M, B as N x K); strides in elements, 0 = shared operand.
 * svgp_spd_inverse_batched: A (batch, m, m) contiguous SPD -> A^-1 in place (blocked Gauss-Jordan, no
 *   pivoting) and logdet[l] = log det A[l]; work holds svgp_spd_inverse_workspace_elems(m, batch) doubles. */
int svgp_dgemm_batched(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                       long long strideA, const double* B, int ldb, long long strideB, double beta, double* C,
                       int ldc, long long strideC, int batch, void* stream);
/* the same GEMM on float64 matrices with the products and sums on the float32 MFMA (operands rounded to float32 while
 * staged into LDS, float32 accumulation; twice the matrix rate): the arithmetic of the reference's float32 SPRITES graph
 * (VAE_utils.py:277, SVGPVAE_model.py:516) on float64 storage; and the float32 GEMM of the float32 networks            */
int svgp_dgemm_f32c_batched(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda,
                            long long strideA, const double* B, int ldb, long long strideB, double beta, double* C,
                            int ldc, long long strideC, int batch, void* stream);
int svgp_sgemm_batched(int ta, int tb, int M, int N, int K, float alpha, const float* A, int lda, long long strideA,
                       const float* B, int ldb, long long strideB, float beta, float* C, int ldc, long long strideC,
                       int batch, void* stream);
/* one GEMM with few output tiles and a long contraction (dense layers / their weight gradients): K is cut into slices
 * that run as a batch into partial products in `scratch` (svgp_dgemm_splitk_scratch_elems doubles, 0 = no split) and
 * are added in fixed order; same operand conventions as svgp_dgemm_batched with batch 1 */
long long svgp_dgemm_splitk_scratch_elems(int M, int N, int K);
int svgp_dgemm_splitk(int ta, int tb, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                      int ldb, double beta, double* C, int ldc, double* scratch, long long scratch_elems, void* stream);
size_t svgp_spd_inverse_workspace_elems(int m, int batch);
int svgp_spd_inverse_batched(int m, int batch, double* A, double* logdet, double* work, void* stream);
/* Cholesky family (cholesky.hip) -- north_star's "Cholesky of K_mm and the triangular solves"; they replace
 * tf.linalg.cholesky + log(diag_part) (SVGPVAE_model.py:270-274) and the solves / inverses behind :239,319,331.
 * Blocked right-looking with 64 x 64 diagonal blocks factored in LDS; panel solve, trailing update and every product of
 * the triangular inverse are the batched f64 MFMA GEMM with tiles / k-panels skipped where an operand is structurally 0.
 * svgp_potrf_batched: A (batch; m x m, row stride lda, batch stride strideA) SPD -> lower factor L in place (strict upper
 *   triangle zeroed), logdet[l] = log det A[l] = 2 sum log diag L.  A non-positive pivot gives NaNs (no error code: no
 *   host synchronisation).  On return the head of `work` holds the inverses of the 64 x 64 diagonal blocks of L,
 *   (batch, ceil(m/64), 64, 64), which svgp_potri_batched accepts as `linv_blocks`.
 * svgp_trsm_batched: side 0: op(L) X = B with B (batch; m x n);  side 1: X op(L) = B with B (batch; n x m);  trans 1:
 *   op(L) = L^T.  L lower triangular (its strict upper part is ignored), strideL = 0: one L for the whole batch.  B <- X.
 * svgp_potri_batched: A (batch, m, m) contiguous holding L -> A^-1 = L^-T L^-1 (full symmetric matrix); linv_blocks
 *   may be NULL (recomputed).  svgp_spd_inverse_batched is potrf + potri for m >= 512.                               */
size_t svgp_potrf_workspace_elems(int m, int batch);
size_t svgp_trsm_workspace_elems(int m, int n, int batch);
size_t svgp_potri_workspace_elems(int m, int batch);
int svgp_potrf_batched(int m, int batch, double* A, int lda, long long strideA, double* logdet, double* work, void* stream);
int svgp_trsm_batched(int side, int trans, int m, int n, const double* L, int ldl, long long strideL, double* B, int ldb,
                      long long strideB, int batch, double* work, void* stream);
int svgp_potri_batched(int m, int batch, double* A, const double* linv_blocks, double* work, void* stream);
/* Inverse of ONE general (m x m, m <= 2048) matrix by LU with partial pivoting (lu.hip): the reference's tf.linalg.inv(K_mm)
 * WITHOUT jitter in the SPRITES conditional-generation path (SPRITES_experiment.py:178; consumed at SVGPVAE_model.py:610-635).
 * With the linear x linear kernels K_mm has rank <= 128 < m = 800, where an elimination without pivoting has no answer.
 * A (row-major, contiguous) is not modified; Ainv must not alias it; work: svgp_lu_inverse_workspace_elems(m) doubles. */
size_t svgp_lu_inverse_workspace_elems(int m);
int svgp_lu_inverse(int m, const double* A, double* Ainv, double* work, void* stream);

/* ---- generic NHWC float64 convolution as a tap-table gather-GEMM on the f64 MFMA (conv_taps.hip) ---------
 * out[n][y*osy+ooy][x*osx+oox][co] = act(bias[co] + sum_t sum_ci in[n][y*sy+oy_t][x*sx+ox_t][ci] W_t[ci][co]),
 * (y,x) in Hs x Ws, reads outside the input are zero; W_t = w + woff[t] is a (Ci x Co) row-major matrix.
 * One descriptor = one class; classes of a launch share n, Hs, Ws.  With suitable tables: Keras Conv2D
 * 3x3/2x2, stride 1/2, same/valid; UpSampling2D(2)+Conv2D as four parity classes on effective weights; all
 * data gradients.  Replaces the Keras layers of spritesVAE / sprites_representation_network
 * (VAE_utils.py:294-338,375-391).  act: 0 none, 1 bias + ELU, 2 bias only.                             */
typedef struct {
    int32_t n, Hi, Wi, Ci;      /* input tensor (n,Hi,Wi,Ci)                                  */
    int32_t Ho, Wo, Co;         /* output tensor (n,Ho,Wo,Co)                                 */
    int32_t Hs, Ws;             /* iteration space of this class                              */
    int32_t sy, sx;             /* input stride per iteration step                            */
    int32_t osy, osx, ooy, oox; /* output placement                                           */
    int32_t nt, act;            /* taps (1..16), activation                                   */
    int32_t oy[16], ox[16], woff[16];
} svgp_conv_desc;
int svgp_conv_taps_fwd(const svgp_conv_desc* d, int ncls, const double* in, const double* w, const double* bias,
                       double* out, void* stream);
/* dW_t[ci][co] = sum_{n,y,x} in[...] * dout[...] for the same tap tables; part: (nwg, part_stride) scratch (the classes' tap ranges must be disjoint);
 * dw (part_stride values, laid out by woff) = fixed-order sum (accumulate != 0 adds to dw).               */
int svgp_conv_taps_wgrad(const svgp_conv_desc* d, int ncls, const double* in, const double* dout, double* part,
                         int nwg, int part_stride, double* dw, int accumulate, void* stream);
/* dpre = dout * elu'(out) in place on dout (out == NULL: no activation) and db[c] = sum over pixels of dpre;
 * part: (1024, C) scratch.                                                                                  */
/* UpSampling2D(2) + Conv2D 3x3 as four parity classes: effective weights we (2,2,2,2,Ci,Co) from w (3,3,Ci,Co), and the
 * gradient of w from the gradient of we (VAE_utils.py:317-338 decoder layers) */
int svgp_upconv_weights(int Ci, int Co, const double* w, double* we, void* stream);
/* Deferred partial sums.  Every svgp_conv_taps_wgrad_fused call ends in two small reductions (weight and bias partials ->
 * dw, db): 32 launches of ~10 us in the launch chains of a SPRITES step, whose results only the optimiser reads.  The *_jobs
 * forms run the convolution kernels as usual but RETURN the reductions as descriptors (at most `cap`, *n_jobs written);
 * svgp_sum_partials_multi[_f32] runs any number of them as one launch (per 64 jobs), adding in the same order as the
 * immediate form.  The caller keeps part / part_b intact until then (one scratch region per layer) and orders `stream`
 * behind every stream that produced partials.                                                                         */
typedef struct {
    const void* part;   /* (ng, stride) partials                     */
    void* out;          /* (len) sums                                */
    int32_t ng, len, stride, accumulate;
} svgp_sum_job;
int svgp_conv_taps_wgrad_fused_jobs(const svgp_conv_desc* d, int ncls, const double* in, const double* out, double* dout,
                                    double* part, double* part_b, int nwg, int part_stride, double* dw, double* db,
                                    svgp_sum_job* jobs, int cap, int* n_jobs, void* stream);
int svgp_conv_taps_wgrad_fused_jobs_f32(const svgp_conv_desc* d, int ncls, const float* in, const float* out, float* dout,
                                        float* part, float* part_b, int nwg, int part_stride, float* dw, float* db,
                                        svgp_sum_job* jobs, int cap, int* n_jobs, void* stream);
int svgp_sum_partials_multi(const svgp_sum_job* jobs, int n, void* stream);
int svgp_sum_partials_multi_f32(const svgp_sum_job* jobs, int n, void* stream);
/* w (nt, A, B) -> wt (nt, B, A): the transposed tap weights a data gradient convolves with; element-type casts of the
 * float32-network engine (master parameters / gradients are float64)                                                 */
int svgp_transpose_taps(int nt, int A, int B, const double* w, double* wt, void* stream);
int svgp_transpose_taps_f32(int nt, int A, int B, const float* w, float* wt, void* stream);
int svgp_cast_f64_f32(long long n, const double* x, float* y, void* stream);
int svgp_cast_f32_f64(long long n, const float* x, double* y, void* stream);
int svgp_upconv_fold_wgrad(int Ci, int Co, const double* ge, double* g, void* stream);
int svgp_elu_bwd_bias(long long npix, int C, const double* out, double* dout, double* part, double* db,
                      void* stream);
/* float32 instantiations of the five convolution entry points above: the reference's dtype for the SPRITES networks
 * (VAE_utils.py:277 `dtype = tf.float32`).  Same descriptors and tap tables; the gather-GEMM runs on
 * v_mfma_f32_16x16x4_f32.                                                                                             */
int svgp_conv_taps_fwd_f32(const svgp_conv_desc* d, int ncls, const float* in, const float* w, const float* bias,
                           float* out, void* stream);
int svgp_conv_taps_wgrad_f32(const svgp_conv_desc* d, int ncls, const float* in, const float* dout, float* part, int nwg,
                             int part_stride, float* dw, int accumulate, void* stream);
int svgp_upconv_weights_f32(int Ci, int Co, const float* w, float* we, void* stream);
/* Reverse pass of a layer's activation, bias and weights in ONE pass over dout (tf.gradients of Conv2D + bias + elu,
 * VAE_utils.py:294-338): dpre = dout * elu'(out) in place on dout (out == NULL: dpre = dout), db[co] = sum of dpre over
 * pixels, dW_t[ci][co] = sum in * dpre (layout by woff).  The classes' output pixel sets must be disjoint and together
 * cover dout.  16 input channels with 4 or 9 taps in every class: one fused kernel (every wave stages its own halo, no
 * workgroup barrier in the pixel loop); any other shape: svgp_elu_bwd_bias + svgp_conv_taps_wgrad in sequence.
 * part_b: (1024, 16) scratch; part: (nwg, part_stride) scratch. */
int svgp_conv_taps_wgrad_fused(const svgp_conv_desc* d, int ncls, const double* in, const double* out, double* dout,
                               double* part, double* part_b, int nwg, int part_stride, double* dw, double* db,
                               void* stream);
int svgp_conv_taps_wgrad_fused_f32(const svgp_conv_desc* d, int ncls, const float* in, const float* out, float* dout,
                                   float* part, float* part_b, int nwg, int part_stride, float* dw, float* db,
                                   void* stream);
int svgp_upconv_fold_wgrad_f32(int Ci, int Co, const float* ge, float* g, void* stream);
int svgp_elu_bwd_bias_f32(long long npix, int C, const float* out, float* dout, float* part, float* db, void* stream);

/* ---- SPRITES pieces (gp_sprites.hip) ----------------------------------------------------------------------
 * spritesSVGP.kernel_matrix (SVGPVAE_model.py:550-600): K = k_action * k_character, each Linear (optionally
 * cosine-normalised) or ExponentiatedQuadratic (k_se; se = [l_action, sigma_action, l_character, sigma_character]).
 * aux (b, 1+Lc) = [action id, character vector]; ip (m, La+Lc); table = GPLVM action vectors (n_act, La).        */
typedef struct {
    int32_t b, m, La, Lc, n_act, normalize, k_se;
    double rep_weight;          /* weight of Kbar: m <= 64: 1 on rank 0 (Kbar is replicated); m > 64: 1 on every rank (the
                                 * reverse factor stage has weighted the replicated part with cfg.rep_weight itself) */
} svgp_sprites_kcfg;
int svgp_sprites_kernel_matrix_fwd(const svgp_sprites_kcfg*, const double* aux, const double* ip, const double* table,
                                   const double* se, double* K, double* Kn, double* knn, void* stream);
/* VJP: d_ip (m,La+Lc), d_table (n_act,La), d_char (b,Lc) (gradient of the batch character vectors), d_se (4).
 * scratch: scratch_elems doubles (row partials of the tiled pass).  svgp_sprites_kernel_bwd_scratch_elems(cfg) reads cfg.b as
 * the row CAPACITY and returns a size that covers every call with 1 <= b <= cfg.b (the need is not monotone in b); the call
 * itself fails with SVGP_ERR_INVALID when scratch_elems is below what its own b needs.                             */
long long svgp_sprites_kernel_bwd_scratch_elems(const svgp_sprites_kcfg*);
int svgp_sprites_kernel_matrix_bwd(const svgp_sprites_kcfg*, const double* aux, const double* ip, const double* table,
                                   const double* se, const double* Kbar, const double* Knbar, const double* knnbar,
                                   double* d_ip, double* d_table, double* d_char, double* d_se, double* scratch,
                                   long long scratch_elems, void* stream);
/* aux_data_SVGPVAE_sprites (SVGPVAE_model.py:1086-1115): segment_mean over seg_len consecutive frames, repeat,
 * prepend the action id; and its reverse.                                                                        */
int svgp_sprites_aux_fwd(int b, int seg_len, int Lc, const double* repr, const double* action_ids, double* aux,
                         void* stream);
int svgp_sprites_aux_bwd(int b, int seg_len, int Lc, const double* d_char, double* d_repr, void* stream);
/* AveragePooling2D over the whole (HW) map of an (n,HW,C) tensor (VAE_utils.py:388) and its reverse */
int svgp_avgpool_fwd(int n, int HW, int C, const double* x, double* y, void* stream);
int svgp_avgpool_bwd(int n, int HW, int C, const double* dy, double* dx, void* stream);
/* encoder head: enc (b,2L) += bias; mu = enc[:, :L], var_raw = exp(enc[:, L:]), var = clip (VAE_utils.py:347-348,
 * SVGPVAE_model.py:858-859); reverse through exp and the clip mask                                             */
int svgp_enc_head_fwd(int b, int L, int clip, const double* bias, double* enc, double* mu, double* var_raw,
                      double* var, void* stream);
int svgp_enc_head_bwd(int b, int L, int clip, const double* var_raw, const double* ybar, const double* s2bar,
                      double* d_enc, void* stream);
int svgp_bias_add(long long rows, int C, const double* bias, double* x, void* stream);
/* utils.py:483-504 gauss_cross_entropy, element-wise over n values: out = -1/2 (log 2 pi + log var2 +
 * (var1 + mu1^2 - 2 mu1 mu2 + mu2^2) / var2).  Inside the training step the same term is evaluated in the
 * per-sample kernels; this is the stand-alone form behind utils.gauss_cross_entropy                          */
int svgp_gauss_cross_entropy(long long n, const double* mu1, const double* var1, const double* mu2, const double* var2,
                             double* out, void* stream);
/* classification loss of the representation-network pre-training (SPRITES_utils.py:335-368): mean sparse softmax
 * cross-entropy over n rows of C logits, labels = class ids as float64; writes per-row losses, their mean and
 * d mean / d logits */
int svgp_softmax_xent(int n, int C, const double* logits, const double* labels, double* row_loss, double* loss,
                      double* dlogits, void* stream);
/* sum (x - xhat)^2 partials into the workspace's partial-sum area (block g writes part_sums[4g+2]) and
 * d loss / d xhat (beta-ELBO: 2(xhat-x)/n_pix; GECO: lagrange_mult/(b_global n_pix) times that)              */
int svgp_sqerr_fwd(long long tot, int n_part, const double* x, const double* xhat, double* part_sums, void* stream);
int svgp_sqerr_bwd(long long tot, int geco, int b_global, int n_pix, const double* state, const double* x,
                   const double* xhat, double* dxhat, void* stream);
/* float32 forms of the per-frame glue of the float32 networks; the squared-error partial sums and the device state stay
 * float64 (they feed the float64 scalar epilogue)                                                                     */
int svgp_avgpool_fwd_f32(int n, int HW, int C, const float* x, float* y, void* stream);
int svgp_avgpool_bwd_f32(int n, int HW, int C, const float* dy, float* dx, void* stream);
int svgp_bias_add_f32(long long rows, int C, const float* bias, float* x, void* stream);
int svgp_sqerr_fwd_f32(long long tot, int n_part, const float* x, const float* xhat, double* part_sums, void* stream);
int svgp_sqerr_bwd_f32(long long tot, int geco, int b_global, int n_pix, const double* state, const float* x,
                       const float* xhat, float* dxhat, void* stream);
/* tf.clip_by_value(grad, -thr, thr) (SPRITES_experiment.py:234-235) */
int svgp_clip_by_value(long long tot, double thr, double* g, void* stream);

/* ---- moving-ball experiment (ball.hip): BALL_experiment.py --elbo SVGPVAE_Hensman | SVGPVAE_Titsias ----------
 * build_SVGPVAE_elbo_graph (SVGPVAE_model.py:638-715) runs its two SVGP objects (:17-171) on the shared GP stage
 * entry points above with rows = the tmax frames, channels = the videos of the batch (all videos share the time
 * stamps 1..tmax, :663-664), N_train = b = tmax, kl_form = 1, clip_pv = 2: one workspace per latent coordinate.
 * svgp_se1d_kernel_matrix_fwd/bwd: tfk.ExponentiatedQuadratic(amplitude=None, length_scale) on scalar inputs (:60,
 *   :80-86): K (m,m), Kn (T,m), knn (T) = 1 from times x (T), inducing points z (m), *ls; VJP -> d_z (m), d_ls (1).
 * svgp_bias_act_fwd / svgp_act_bwd_bias: x = act(x + bias) in place (act 0 none, 1 tanh); dpre = dout act'(out) in
 *   place and db = column sums (one launch, fixed summation order; `part` is unused and may be NULL) - the layers of
 *   build_MLP_inference_graph / build_MLP_decoder_graph (VAE_utils.py:9-96); the matmuls are svgp_dgemm_batched.
 * svgp_ball_head_fwd/bwd: h (B*T,4) + bias -> qnet_mu, exp, clip [1e-6,1e3] (VAE_utils.py:50-55, SVGPVAE_model.py:
 *   670-671) written in the (T,B) channel layout of the x / y workspaces; reverse through exp and the clip mask.
 * svgp_ball_pack_z / unpack_zbar: latent samples (T,B) x 2 <-> (B*T,2).
 * svgp_sigmoid_xent: per frame -sum_pix sigmoid_cross_entropy_with_logits (:697-700), pred = sigmoid, and
 *   dlogits = scale (sigmoid - label); pred / dlogits may be NULL.
 * svgp_ball_elbo_assemble: per-video [elbo, recon, KL_term, inside_elbo, ce_term, inside_recon, inside_kl] (7,B)
 *   (:677-705), incl. the reference's batch-wide KL scalar; svgp_ball_finalize: their means over videos into the
 *   state's output slots (BALL_experiment.py:116-123), Adam step counter and RNG counter advance.               */
int svgp_se1d_kernel_matrix_fwd(int T, int m, const double* x, const double* z, const double* ls, double* K,
                                double* Kn, double* knn, void* stream);
int svgp_se1d_kernel_matrix_bwd(int T, int m, const double* x, const double* z, const double* ls, const double* Kbar,
                                const double* Knbar, double* d_z, double* d_ls, void* stream);
int svgp_bias_act_fwd(long long rows, int C, int act, const double* bias, double* x, void* stream);
int svgp_act_bwd_bias_scratch_elems(int C);
int svgp_act_bwd_bias(int rows, int C, int act, const double* out, double* dout, double* part, double* db,
                      void* stream);
int svgp_ball_head_fwd(int B, int T, int clip, const double* bias, const double* h, double* mu_x, double* var_raw_x,
                       double* var_x, double* mu_y, double* var_raw_y, double* var_y, void* stream);
int svgp_ball_head_bwd(int B, int T, int clip, const double* var_raw_x, const double* ybar_x, const double* s2bar_x,
                       const double* var_raw_y, const double* ybar_y, const double* s2bar_y, double* dh, void* stream);
int svgp_ball_pack_z(int B, int T, const double* zx, const double* zy, double* z, void* stream);
int svgp_ball_unpack_zbar(int B, int T, const double* dz, double* zbar_x, double* zbar_y, void* stream);
int svgp_sigmoid_xent(int rows, int P, double scale, const double* logits, const double* labels, double* pred,
                      double* row_recon, double* dlogits, void* stream);
int svgp_ball_elbo_assemble(const svgp_mnist_cfg* cfg_x, const double* ws_x, const double* ws_y,
                            const double* row_recon, const double* state, double* out, void* stream);
int svgp_ball_finalize(int B, int did_adam, long long rng_advance, const double* out, double* state, void* stream);
int svgp_state_add(double* state, int slot, double v, void* stream);
/* build_video_batch_graph's rasterisation (utils.py:172-187): vid[f][i][j] = (i - paths[f][0])^2 + (j - paths[f][1])^2 < r^2 */
int svgp_ball_rasterize(long long frames, int px, int py, double r, const double* paths, double* vid, void* stream);

/* ---- Pearce baseline of the moving-ball experiment: BALL_experiment.py --elbo GPVAE_Pearce | VAE | NP ----------
 * build_1d_gp (GPVAE_Pearce_model.py:8-86) with X_test = X: exact GP regression per (video, latent coordinate),
 * one workgroup each, n <= 64 points:  A = K_SE(l) + diag(var), alpha = A^-1 y,
 *   lhood = -1/2 (n log 2pi + y.alpha + log det A),  p_m = K alpha,  p_v = 1 - diag(K A^-1 K),
 *   z = p_m + eps sqrt(p_v), per-frame -gauss_cross_entropy (utils.py:483-504) and their sums.
 * All per-frame buffers are (T, B) (frame-major, the layout svgp_ball_head_fwd writes); times (T).
 * idx (B, n) int32 != NULL: the points are times[idx[b][.]] (neural-process context sets, :121-155): forward gives
 * lhood only, reverse seeds it with seed_lh_scale (-1: the context likelihood is subtracted, :186) and accumulates.
 * tmask (B, T): weights of the cross-entropy terms (NP: 1 on target frames, :178-182).
 * svgp_pearce_elbo_assemble: per-video [elbo, recon, prior_kl, lhood, ce, context lhood, 0] (7,B) (:157-236). */
typedef struct {
    int32_t B, T, n;
    const double* times; const int32_t* idx; const double* tmask;
    const double* ls_x; const double* ls_y;
    double* y_x; double* y_y; double* s2_x; double* s2_y;
    double* p_m_x; double* p_m_y; double* p_v_x; double* p_v_y; double* eps_x; double* eps_y; double* z_x; double* z_y;
    double* zbar_x; double* zbar_y; double* ybar_x; double* ybar_y; double* s2bar_x; double* s2bar_y;
    double* Ai; double* alpha; double* lh; double* ce; double* row_ce; double* dl_part;
} svgp_pearce_bufs;
int svgp_pearce_gp_fwd(const svgp_pearce_bufs*, const double* eps_x, const double* eps_y, const double* state,
                       void* stream);
int svgp_pearce_gp_bwd(const svgp_pearce_bufs*, double seed_lh_scale, int accumulate, const double* state,
                       double* d_ls_x, double* d_ls_y, void* stream);
int svgp_pearce_elbo_assemble(int B, int T, const double* lh, const double* ce, const double* con_lh,
                              const double* row_recon, const double* row_ce, const double* tmask, const double* state,
                              double* out, void* stream);
int svgp_scale_rows(long long rows, int C, const double* w, double* x, void* stream);

/* ---- deep SVIGP_Hensman baseline on rotated MNIST (svigp.hip): MNIST_experiment.py --elbo SVIGP_Hensman -------------
 * SVIGP_Hensman.variational_loss / approximate_posterior_params (SVIGP_Hensman_model.py:135-227) and the ELBO assembly
 * of forward_pass_deep_SVIGP_Hensman (:230-289) with free variational parameters loc (L,m), scale (L,m,m)
 * (S_l = scale_l scale_l^T) and a likelihood noise (1); kernel matrices from svgp_kernel_matrix_fwd, decoder =
 * svgp_mnist_decoder_fwd/bwd.  svgp_svigp_fwd: Z (b,L) = K_nm (K_mm + jI)^-1 loc_l (training mean vectors, and with
 * test-point kernel matrices the prediction of :292-339).  svgp_svigp_bwd: reverse pass of -elbo given the decoder's
 * Zbar (beta-ELBO seed, scaled here by n_pix / (2 noise^2)) and its squared-error partial sums; writes Kbar / Knbar /
 * knnbar for svgp_kernel_matrix_bwd and the gradients of loc, scale, noise.  svgp_svigp_assemble: out (7) = [elbo,
 * recon_loss / n_pix, KL_term, inside_elbo, 0, inside_recon, inside_kl].  ws: svgp_svigp_workspace_elems doubles;
 * ws[svgp_svigp_scale_offset] holds n_pix / (2 noise^2), the factor the decoder's weight gradients take
 * (svgp_scale_by_device_scalar).                                                                                    */
long long svgp_svigp_workspace_elems(int b, int m, int L);
long long svgp_svigp_scale_offset(int b, int m, int L);
int svgp_svigp_fwd(int b, int b_global, int m, int L, int n_pix, double jitter, const double* K, const double* Kn,
                   const double* knn, const double* loc, const double* scale, const double* noise, double* Z,
                   double* ws, void* stream);
int svgp_svigp_bwd(int b, int b_global, int m, int L, int n_pix, double N_train, const double* Kn, const double* loc,
                   const double* scale, const double* noise, double* Zbar, const double* part_sums, int n_part,
                   double* Kbar, double* Knbar, double* knnbar, double* d_loc, double* d_scale, double* d_noise,
                   double* ws, void* stream);
int svgp_svigp_assemble(int b, int b_global, int m, int L, int n_pix, double N_train, const double* noise,
                        const double* part_sums, int n_part, const double* ws, double* out, void* stream);
int svgp_scale_by_device_scalar(long long n, const double* f, double* x, void* stream);

/* ---- runtime helpers: HIP graphs and events without going through torch ---------------------*/
int svgp_stream_create(void** stream_out);
/* HIP maps streams onto a small pool of hardware queues (GPU_MAX_HW_QUEUES) in creation order; two streams on one queue run
 * one after the other whatever the events say.  svgp_streams_overlap: *out = 1 iff a kernel on b runs BESIDE a kernel on a
 * (a ~40 us spin on a, an empty kernel on b; synchronises both).  svgp_side_streams_prepare: creates the library's two side
 * branches of `stream` now -- picked with that probe from up to 12 candidates -- instead of at the first step (which may
 * be under stream capture, where the probe cannot run).  SVGP_STREAM_PROBE=0 disables the probe.                          */
int svgp_streams_overlap(void* a, void* b, int* out);
int svgp_side_streams_prepare(void* stream);
int svgp_stream_destroy(void* stream);
int svgp_stream_sync(void* stream);
int svgp_graph_begin(void* stream);                       /* hipStreamBeginCapture               */
int svgp_graph_end(void* stream, void** graph_exec_out);  /* EndCapture + Instantiate            */
int svgp_graph_launch(void* graph_exec, void* stream);
int svgp_graph_destroy(void* graph_exec);
int svgp_event_create(void** event_out);
int svgp_event_record(void* event, void* stream);
int svgp_event_elapsed_ms(void* start, void* stop, float* ms_out); /* synchronises on `stop`      */
int svgp_event_destroy(void* event);

#ifdef __cplusplus
}
#endif
#endif /* SVGPVAE_HIP_H */
