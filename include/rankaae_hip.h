/* rankaae_hip.h -- C ABI of librankaae_hip.so (MI355X / gfx950 only).
 *
 * The reference (AI-multimodal/RankAAE) has no FFI: its hot path is eager PyTorch.
 * Each entry point below replaces a chain of ATen ops *plus its autograd backward*
 * in the reference file:line cited.  Conventions:
 *   - every pointer is a DEVICE pointer (fp32 unless said otherwise); no allocation,
 *     no host synchronisation inside; work is enqueued on `stream` (a hipStream_t);
 *   - return value: 0 on success, a hipError_t (>0) from the launch, or
 *     RAAE_EINVAL (-1) when host-side shape validation fails (nothing is launched);
 *   - reductions that cross workgroups use fixed-order partial buffers, never float
 *     atomics, so every result is bitwise reproducible run to run.
 *
 * Batch-norm convention ("raw + partials"): a producing kernel stores the RAW tensor
 * and per-workgroup partial sums {sum, sum of squares} (double) of what the following
 * BatchNorm1d(affine=False) sees; the consuming kernel reduces those partials in its
 * prologue (train mode, biased variance, eps) or takes running statistics (eval mode).
 */
#ifndef RANKAAE_HIP_H
#define RANKAAE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RAAE_EINVAL (-1)
#define RAAE_MAX_PARTS 512      /* max partial-sum rows any producer may emit */

/* Source of BatchNorm statistics for a consumer (torch.nn.BatchNorm1d(affine=False),
 * sc/clustering/model.py:29,35,49,110,116,130,250,284,349,358,366,460,543,552). */
typedef struct {
    const double* partials;   /* [nparts][C][2] {sum, sumsq}; NULL => eval mode (use running stats) */
    int nparts;
    float count;              /* elements per channel behind the partials (B*L) */
    float* running_mean;      /* [C]; read in eval mode; updated in train mode iff update_running */
    float* running_var;       /* [C] */
    float momentum;           /* 0.1 */
    float eps;                /* 1e-5 */
    int update_running;       /* exactly one consumer of a BN layer per forward sets this */
} raae_bn_t;

/* Dropout multipliers generated inside the consuming kernel (rng_mode "philox", no tape read): element e of the slot is
 * kept iff hash(e + offset; k1, k2) < thr and then scales by inv = 1/keep -- the same function raae_rng_fill evaluates
 * for tape-resident slots, so both forms of a slot are bit-identical (nn.Dropout of the reference draws from the global
 * CPU generator instead: sc/clustering/model.py:337,351,360,526,545; parity mode keeps the host tape).  (k1, k2) are
 * the step's hash keys, a function of (seed, step counter) that raae_step_begin / raae_step_tick store next to the
 * counter once per step.  keys == NULL: disabled (the consumer reads its `mask` tensor, or applies none). */
typedef struct {
    const unsigned* keys;              /* device: {k1, k2} of the current step (engine: rng_state + 2 words of 8 bytes) */
    unsigned offset;                   /* position of the slot's element 0 in the numbering of all dropout elements of a step */
    unsigned thr;                      /* min(2^32 - 1, (double)(float)keep * 2^32) */
    float inv;                         /* 1.f / (float)keep */
} raae_maskgen_t;

/* ---- input transform applied while loading a dense layer's input ---- */
enum { RAAE_IN_NONE = 0, RAAE_IN_PRELU_BN_DROP = 1, RAAE_IN_PRELU_DROP = 2 };
/* ---- what the dense layer's output feeds ---- */
enum { RAAE_OUT_RAW = 0,          /* store z; no statistics */
       RAAE_OUT_STATS_PRELU = 1,  /* store z; partials of PReLU(z)   (Linear->PReLU->BN) */
       RAAE_OUT_STATS_RAW = 2,    /* store z; partials of z          (Linear->BN)        */
       RAAE_OUT_SOFTPLUS = 3,     /* store softplus_beta2(z)         (decoder output)    */
       RAAE_OUT_RELU = 4 };       /* store relu(z) */

/* Fused  [PReLU -> BatchNorm -> Dropout] -> Linear  forward on the matrix cores
 * (v_mfma_f32_16x16x4_f32: exact fp32).  Replaces nn.Linear + the preceding
 * PReLU/BatchNorm1d/Dropout modules of FCEncoder / FCDecoder / DiscriminatorFC
 * (sc/clustering/model.py:346-371, 540-563, 635-653) and lin3 (model.py:283).
 *   x      [B][K]  raw output of the previous layer (or the network input)
 *   slope  [K]     PReLU slopes of the previous layer      (in_kind != NONE)
 *   bn             statistics of PReLU(x)                  (in_kind == PRELU_BN_DROP)
 *   mask   [B][K]  dropout scale {0, 1/(1-p)} or NULL
 *   w [N][K], bias [N]; z [B][N] output
 *   out_slope [N]  PReLU slopes of THIS layer (out_kind == STATS_PRELU)
 *   out_partials   [grid_x][N][2] doubles; *out_nparts receives grid_x (host int)
 */
int raae_dense_fwd(const float* x, int B, int K, int in_kind, const float* slope, const raae_bn_t* bn,
                   const float* mask, const float* w, const float* bias, int N, float* z, int out_kind,
                   const float* out_slope, double* out_partials, int* out_nparts, void* stream);

/* Two raae_dense_fwd calls that do not depend on each other (a layer of the encoder and a layer of the decoder,
 * while the forward chain the reference discards runs beside one that is needed) in ONE launch.  The struct holds
 * raae_dense_fwd's arguments; has_bn = 0 stands for bn == NULL. */
typedef struct {
    const float* x; int B, K, in_kind; const float* slope; int has_bn; raae_bn_t bn; const float* mask;
    const float* w; const float* bias; int N; float* z; int out_kind; const float* out_slope; double* out_partials;
    int storage;          /* RAAE_ST_* bits: which tensors are stored as bf16 (0: all fp32) */
    float mask_scale;     /* RAAE_ST_MASK: the bf16 mask holds {0, 1} and is multiplied by this fp32 1/(1-p) (a bf16
                             1/(1-p) would bias every activation by -0.16 %); 0 is read as 1 */
    raae_maskgen_t gen;   /* gen.keys != NULL: dropout multipliers generated in the kernel, `mask` must be NULL */
} raae_dense_fwd_t;
int raae_dense_fwd2(const raae_dense_fwd_t* p, const raae_dense_fwd_t* q, int* nparts_p, int* nparts_q, void* stream);

/* bf16 STORAGE of activations and dropout multipliers (build-only config key `precision: bf16`, BASELINE configs[4]:
 * 512-point spectra + 12 descriptors, mixed precision).  The reference arithmetic is fp32
 * (sc/clustering/dataloader.py:61); this mode halves the bytes of the [B][hidden] activations and masks while every
 * product, sum, BatchNorm statistic, gradient and Adam moment stays fp32 / double.  A set bit means the tensor behind
 * the (still float*-typed) pointer holds bf16 values; the layer output is rounded (nearest even) before its
 * BatchNorm statistics are taken.  storage == 0 is the fp32 path, instruction for instruction. */
enum { RAAE_ST_X = 1,            /* layer input x                    */
       RAAE_ST_MASK = 2,         /* dropout multipliers of the input */
       RAAE_ST_Z = 4 };          /* raw layer output z (forward) / zout (backward) */
/* raae_dense_fwd from a raae_dense_fwd_t, honouring `storage` */
int raae_dense_fwd_s(const raae_dense_fwd_t* p, int* out_nparts, void* stream);

/* how the gradient w.r.t. this layer's raw output z is obtained in the prologue */
enum { RAAE_G_DIRECT = 0,        /* g is dL/dz                                                  */
       RAAE_G_SOFTPLUS = 1,      /* g is dL/d softplus(z); `zout` holds softplus(z)             */
       RAAE_G_PRELU_BN = 2,      /* g is dL/dy, y = BN(PReLU(z)); needs g_partials, z, bn, slope */
       RAAE_G_PRELU = 3,         /* g is dL/d PReLU(z) (no BN: discriminator)                   */
       RAAE_G_RELU = 4 };

/* Backward of the same fused layer: dW, dbias, dslope (fixed-order slabs, one per
 * workgroup), and -- if dx != NULL -- dL/d(input after its transform's BatchNorm),
 * i.e. dx = (g_z . W) * mask, together with the partial sums {sum dx, sum dx*y_in}
 * that the previous layer's RAAE_G_PRELU_BN prologue needs (autograd of
 * nn.Linear/PReLU/BatchNorm1d/Dropout; torch semantics).
 *   g [B][N], g_partials [g_nparts][N][2] ({sum g, sum g*y}), zout [B][N] = this layer's stored output
 *   out_slope [N], out_bn: this layer's own PReLU/BN (for the prologue)
 *   x, in_kind, slope, bn, mask: as in forward (recomputes the layer input)
 *   dw_slab [nslab][slab_stride] base pointers already offset to this tensor:
 *       dW at dw, dbias at db, dslope at dslope (may be NULL when out has no PReLU)
 *   dx [B][K] or NULL; dx_partials [grid][K][2] or NULL (needed iff in_kind==PRELU_BN_DROP)
 *   *nslab receives the number of slabs written (= grid size).
 */
int raae_dense_bwd(const float* g, int g_kind, const double* g_partials, int g_nparts, const float* zout,
                   const float* out_slope, const raae_bn_t* out_bn, int B, int N,
                   const float* x, int K, int in_kind, const float* slope, const raae_bn_t* bn,
                   const float* mask, const float* w,
                   float* dw, float* db, float* dslope, long slab_stride, int* nslab,
                   float* dx, double* dx_partials, void* stream);
/* the same with bf16 storage bits (RAAE_ST_X: x, RAAE_ST_MASK: mask, RAAE_ST_Z: zout) */
int raae_dense_bwd_st(const float* g, int g_kind, const double* g_partials, int g_nparts, const float* zout,
                   const float* out_slope, const raae_bn_t* out_bn, int B, int N,
                   const float* x, int K, int in_kind, const float* slope, const raae_bn_t* bn,
                   const float* mask, const float* w,
                   float* dw, float* db, float* dslope, long slab_stride, int* nslab,
                   float* dx, double* dx_partials, int storage, void* stream);

/* struct form (the arguments of raae_dense_bwd_st + in-kernel dropout multipliers) */
typedef struct {
    const float* g; int g_kind; const double* g_partials; int g_nparts; const float* zout; const float* out_slope;
    int has_out_bn; raae_bn_t out_bn; int B, N;
    const float* x; int K, in_kind; const float* slope; int has_bn; raae_bn_t bn; const float* mask; const float* w;
    float* dw; float* db; float* dslope; long slab_stride; float* dx; double* dx_partials;
    int storage; float mask_scale; raae_maskgen_t gen;
} raae_dense_bwd_t;
int raae_dense_bwd_s(const raae_dense_bwd_t* p, int* nslab, void* stream);

/* Experiment (build-only key `collapse_stats`, default off): the partial rows [n][C][2] of one or
 * two BatchNorm statistics are added once into a single row [1][C][2] (one workgroup each), so that every consumer's
 * prologue reads one row.  p2 == NULL: one statistic. */
int raae_stat_collapse2(const double* p1, int n1, int C1, double* o1, const double* p2, int n2, int C2, double* o2,
                        void* stream);

/* Final BatchNorm1d(nstyle, affine=False) of both encoders (model.py:284,366):
 * styles = BN(z).  Backward: dz from dstyles (torch batch_norm backward, train mode). */
int raae_style_bn_fwd(const float* z, int B, int C, const raae_bn_t* bn, float* styles, void* stream);
int raae_style_bn_bwd(const float* dstyles, const float* styles, int B, int C, const raae_bn_t* bn,
                      float scale, float* dz, void* stream);

/* Pairwise sign-concordance ("Kendall") loss and its gradient in ONE pass over the
 * B^2 pairs, never materialising [B,B,n_aux] (sc/utils/functions.py:37-79 + autograd).
 *   d [B][ldd] descriptors, z [B][ldz] styles (first n_aux columns are used)
 *   work: >= raae_rank_loss_work_bytes(B, n_aux) bytes of scratch
 *   loss: 1 float; dz [B][ldz] (columns >= n_aux are zeroed) or NULL (validation). */
long raae_rank_loss_work_bytes(int B, int n_aux);
int raae_rank_loss_fwd_bwd(const float* d, int ldd, const float* z, int ldz, int B, int n_aux, int activate,
                           void* work, float* loss, float* dz, void* stream);

/* The same loss over the GLOBAL pairs of a data-parallel batch (what sc/utils/functions.py:63-77 computes on the
 * whole batch in one process), split per rank: this rank owns rows [row0, row0 + nrows) of the all-gathered
 * d_all / z_all [n_all][ld] and pairs them with all n_all rows.
 *   raae_rank_rows_pairs:  the pair pass; totals[64] doubles = this rank's {n+[16], n-[16], S+[16], S-[16]}
 *   (the caller sums `totals` over the ranks -- one 512-byte all-reduce)
 *   raae_rank_rows_finish: global loss (identical on every rank) and dz [nrows][ldz] = scale * dL/dz of this
 *   rank's rows (exact: every pair that contains row i is seen by its owner); scale = number of ranks when the
 *   parameter gradients are AVERAGED over ranks afterwards.
 *   work: >= raae_rank_loss_work_bytes(nrows, n_aux) bytes, the same buffer for both calls. */
int raae_rank_rows_pairs(const float* d_all, int ldd, const float* z_all, int ldz, int n_all, int row0, int nrows,
                         int n_aux, void* work, double* totals, void* stream);
int raae_rank_rows_finish(const double* totals, int n_all, int nrows, int n_aux, int activate, float scale,
                          void* work, float* loss, float* dz, int ldz, void* stream);

/* Model-selection metrics of the validation styles (sc/clustering/trainer.py:286-292: scipy.stats.shapiro on
 * every style column, scipy.stats.spearmanr on every column pair of the host copy), computed where the styles
 * already are.  z [n][k] styles; a_coef: the n/2 Shapiro-Wilk coefficients for this n (Royston AS R94 with the
 * AS 111 normal quantile, as scipy 1.15.3 forms them -- the host computes them once per n,
 * rankaae_amd/metrics.py::shapiro_coefficients); work: 2*k*n doubles (average ranks, sorted columns);
 * out: k doubles W_c, then k(k-1)/2 doubles rho_(p,q), p < q, in itertools.combinations order. */
int raae_style_metrics(const float* z, int n, int k, const double* a_coef, double* work, double* out, void* stream);

/* out[g][l] = mean over s of x[g][s][l], x [groups][per][L]: the n_sampling average of the report's decoder
 * sweeps (sc/report/analysis.py:78-86, `decoder(con_c).reshape(n_spec, n_sampling, L).mean(axis=1)`). */
int raae_group_mean(const float* x, int groups, int per, int L, float* out, void* stream);

/* The adversarial branch of a step in ONE launch (DiscriminatorFC with three layers of width `hidden` = 64 and
 * nstyle <= 16; sc/clustering/model.py:631-663, sc/utils/functions.py:109-132, model.py:8-22): input = [z_real ;
 * styles] (+ sigma * noise), Linear/PReLU/Dropout x2, Linear, BCE-with-logits against ones (real rows) and zeros (fake
 * rows), backward with all parameter gradients as *nslab slabs (<= 256), dstyles = -alpha[0] * dL/d(styles).
 * Replaces raae_disc_input + 3 raae_dense_fwd + raae_bce_pair_fwd_bwd + 3 raae_dense_bwd + raae_scale_by_dev.
 * mask1/mask2: dropout multipliers [n][hidden] or NULL; noise [n][ns] or NULL; partial: >= 256 doubles; ticket: one
 * zero-initialised unsigned owned by the caller (the kernel leaves it at zero); loss: 1 float. */
typedef struct {
    const float* z_real; const float* styles; const float* noise; float sigma;
    const float* mask1; const float* mask2;
    const float* w1; const float* b1; const float* s1; const float* w2; const float* b2; const float* s2;
    const float* w3; const float* b3; const float* alpha;
    int n_real, n_fake, ns, hidden;
    float* dw1; float* db1; float* ds1; float* dw2; float* db2; float* ds2; float* dw3; float* db3;
    long slab_stride;
    float* dstyles; double* partial; unsigned* ticket; float* loss;
} raae_disc_fused_t;
int raae_disc_fused(const raae_disc_fused_t* a, int* nslab, void* stream);

/* Optional last argument of the three loss kernels below: finish the loss inside the kernel (the last workgroup to
 * arrive adds the partials in index order: out[slot] = scale * sum, out[acc_slot] += it when acc_slot >= 0) instead of
 * a raae_loss_finalize launch.  ticket: one zero-initialised unsigned owned by the caller, left at zero; NULL (or
 * fin == NULL): the partials are left to the caller. */
typedef struct { float scale; float* out; int slot; int acc_slot; unsigned* ticket; } raae_loss_fin_t;

/* recon_loss (functions.py:81-107): scale!=0 => "flexible target" branch.
 * partial: [>= grid] doubles (fixed-order loss partials); *nparts = grid. dout may be NULL. */
int raae_recon_loss_fwd_bwd(const float* spec_in, const float* spec_out, int B, int L, int scale,
                            double* partial, int* nparts, float* dout, const raae_loss_fin_t* fin, void* stream);

/* smoothness_loss (functions.py:194-212, model.py:177-229): replicate pad, `ntaps`-tap
 * normalised Gaussian (taps given by the host), MSE(x, G x); gradient through both operands. */
int raae_smooth_loss_fwd_bwd(const float* x, int B, int L, const float* taps, int ntaps,
                             double* partial, int* nparts, float* dx, const raae_loss_fin_t* fin, void* stream);

/* nn.MSELoss (mutual_info_loss, functions.py:187-190): mean((a-b)^2), da = 2(a-b)/n. */
int raae_mse_fwd_bwd(const float* a, const float* b, long n, double* partial, int* nparts, float* da,
                     const raae_loss_fin_t* fin, void* stream);

/* BCEWithLogitsLoss(real,1) + BCEWithLogitsLoss(fake,0), each mean-reduced over its
 * own count (functions.py:119-130).  logits [n_real + n_fake]; dlogits same shape. */
int raae_bce_pair_fwd_bwd(const float* logits, int n_real, int n_fake, float* loss, float* dlogits, void* stream);

/* Discriminator input: rows [0,n_real) = z_real + sigma*noise, rows [n_real, n_real+n_fake) =
 * styles + sigma*noise (model.py:658-661; noise==NULL in eval mode). */
int raae_disc_input(const float* z_real, const float* styles, const float* noise, float sigma,
                    int n_real, int n_fake, int C, float* out, void* stream);
/* dst[i] = scale * src[i]  (gradient reversal: scale = -alpha, model.py:15-22; alpha read from device) */
int raae_scale_by_dev(const float* src, const float* dev_scale, float sign, long n, float* dst, void* stream);

/* sums fixed-order partial buffers into loss slots: out[slot] = scale * sum(partial[0..n)),
 * optionally accumulating out[acc_slot] += out[slot] (avg_mutual_info, trainer.py:186). */
int raae_loss_finalize(const double* partial, int n, float scale, float* out, int slot, int acc_slot, void* stream);

/* Batch assembly from the device-resident dataset: rows idx[cursor-B .. cursor) (cursor: device int,
 * already advanced past this batch by raae_step_tick; NULL => idx[0..B)) -> spec_out, aux_out;
 * spec_out += noise * spec_noise (dataloader.py:46-61 + trainer.py:112). */
int raae_gather_batch(const float* spec, const float* aux, const long* idx, const int* cursor, const float* noise,
                      float spec_noise, int B, int L, int n_aux, float* spec_out, float* aux_out, void* stream);

/* Fused multi-tensor Adam/AdamW over a flat arena (torch.optim.Adam/AdamW single-tensor
 * formulas, operation order mirrored in fp32; trainer.py:333-397).  Gradient of element i =
 * fixed-order sum of seg_nslab[i/64] slabs; 0 slabs => skipped (like a param with .grad None).
 *   p, m, v: base pointers of the optimizer's contiguous arena range, n floats (multiple of 64)
 *   g_slabs: slab 0 of the same range; slab s at g_slabs + s*slab_stride
 *   hyper (device, 5 doubles): {lr, beta1, beta2, eps, weight_decay}
 *   step  (device int): 1-based step count, advanced by raae_step_tick BEFORE this launch
 *   decoupled=1 => AdamW (p *= 1-lr*wd), 0 => Adam (g += wd*p)
 *   max_nslab: upper bound of seg_nslab over the range (host-side hint: above 16 the slabs of an element are
 *   summed by 8 lanes instead of one thread; the result does not depend on it beyond summation order). */
int raae_adam_step(float* p, float* m, float* v, const float* g_slabs, long slab_stride, const unsigned short* seg_nslab,
                   long n, const double* hyper, const int* step, int decoupled, int max_nslab, void* stream);
/* ====================== 1-D convolutional networks (ae_form: compact) ======================
 * Activations are [B][C][L] fp32, stored RAW (pre-activation); what a consumer sees is a *view*:
 *     value = mask * BatchNorm( PReLU(raw, slope_c) )          (each stage optional)
 * so PReLU, BatchNorm(affine=False) and Dropout never make a pass over memory of their own. */
typedef struct {
    const float* raw;       /* [B][C][L] */
    const float* slope;     /* [C] PReLU slopes or NULL */
    raae_bn_t bn;           /* statistics over (B, L) per channel, valid iff has_bn */
    int has_bn;
    const float* mask;      /* [B][C][L] dropout scale or NULL */
} raae_view_t;

/* How dL/d(raw output) of a layer is obtained from what its consumer left behind:
 *   dA   = has_bn ? rstd_c * (g - mean_c(g) - y * mean_c(g*y)) : g,   y = (u - mu_c) * rstd_c
 *   dRaw = dA * PReLU'(raw)     (slope != NULL)   |  dA * (1 - exp(-2 out))  (act == softplus, raw = out)
 * `u` is the tensor the BatchNorm normalised: NULL means u = PReLU(raw) (own output), else e.g. the
 * three-way block sum Y.  dslope_c = sum dA * raw [raw <= 0]. */
typedef struct {
    const float* g;               /* [B][C][L] */
    const double* g_partials;     /* [g_nparts][C][2] {sum g, sum g*y}; needed iff has_bn */
    int g_nparts;
    const float* u;
    raae_bn_t bn;
    int has_bn;
    const float* raw;
    const float* slope;
    int act;                      /* RAAE_OUT_RAW | RAAE_OUT_SOFTPLUS | RAAE_OUT_RELU */
} raae_grad_t;

typedef struct {
    int Cin, Lin, Cout, Lout, K, stride, pad, pad_replicate, groups, transposed;
} raae_conv_t;   /* transposed => ConvTranspose1d with K == stride, pad 0 (the only form the reference uses) */

/* nn.Conv1d / nn.ConvTranspose1d forward over a view (EncodingBlock / DecodingBlock convs, reference
 * sc/clustering/model.py:33-38,51,59,114-119,133,140,461).  Weight layouts are torch's:
 * Conv1d [Cout][Cin/groups][K], ConvTranspose1d [Cin][Cout/groups][K].
 * stats_kind: RAAE_OUT_RAW none | RAAE_OUT_STATS_PRELU of PReLU(out,out_slope) | RAAE_OUT_STATS_RAW;
 * act: RAAE_OUT_RAW | RAAE_OUT_SOFTPLUS | RAAE_OUT_RELU applied to what is stored. */
int raae_conv_fwd(const raae_view_t* in, int B, const raae_conv_t* cv, const float* w, const float* bias, float* out,
                  int stats_kind, const float* out_slope, double* out_partials, int* out_nparts, int act, void* stream);
/* dL/d(view value of the input) (= dValue * mask, i.e. w.r.t. the BatchNorm output).  accumulate != 0 adds
 * to din; din_partials (may be NULL) receives {sum din, sum din*y_in} of the FINAL din (after accumulation). */
int raae_conv_bwd_data(const raae_grad_t* go, int B, const raae_conv_t* cv, const float* w, const raae_view_t* in,
                       float* din, int accumulate, double* din_partials, int* din_nparts, void* stream);
/* parameter gradients: dw, dbias (same layouts as w/bias) and dslope [Cout] (may be NULL) as *nslab
 * fixed-order slabs (slab s at ptr + s*slab_stride, *nslab <= RAAE_MAX_PARTS: size the buffer for that), summed
 * by raae_adam_step. */
int raae_conv_bwd_weight(const raae_grad_t* go, int B, const raae_conv_t* cv, const raae_view_t* in,
                         float* dw, float* dbias, float* dslope, long slab_stride, int* nslab, void* stream);

/* The decoder's head backward (reference sc/clustering/model.py:461, BatchNorm1d(C, affine=False) -> Conv1d(C, 1, 1)
 * -> Softplus / ReLU, and its autograd) in ONE streaming pass instead of raae_conv_bwd_data + raae_conv_bwd_weight:
 * din [B][C][L] = dL/d(BatchNorm output), din_partials {sum din, sum din*y} per workgroup (*din_nparts rows), and the
 * parameter-gradient slabs dw [C], dbias [1] (*nslab of them).  raae_head_bwd_supported() != 0 says whether the shape
 * is one this kernel takes (K = 1, Cout = 1, Cin 4 or 8, rows of whole quads, a plain BatchNorm view, `go` without a
 * BatchNorm of its own); raae_conv_fwd runs the matching forward kernel for the same shapes by itself. */
int raae_head_bwd_supported(const raae_grad_t* go, int B, const raae_conv_t* cv, const raae_view_t* in);
int raae_head_bwd(const raae_grad_t* go, int B, const raae_conv_t* cv, const float* w, const raae_view_t* in,
                  float* din, double* din_partials, int* din_nparts, float* dw, float* dbias, long slab_stride,
                  int* nslab, void* stream);

/* nn.Linear applied along the LENGTH axis of [B][C][Lin] -> [B][C][E] (excitation fc1 / fc2, reference
 * model.py:44-47,89-93,125-128,164-167); out_slope / statistics are per CHANNEL c (PReLU on dim 1). */
int raae_lenlin_fwd(const raae_view_t* in, int B, int C, int Lin, const float* w, const float* bias, int E, float* out,
                    int stats_kind, const float* out_slope, double* out_partials, int* out_nparts, void* stream);
int raae_lenlin_bwd_data(const raae_grad_t* go, int B, int C, int E, const float* w, const raae_view_t* in, int Lin,
                         float* din, int accumulate, double* din_partials, int* din_nparts, void* stream);
int raae_lenlin_bwd_weight(const raae_grad_t* go, int B, int C, int E, const raae_view_t* in, int Lin,
                           float* dw, float* dbias, float* dslope, long slab_stride, int* nslab, void* stream);

/* Block output: y = view_a + view_b + view_c (reference model.py:99,173), statistics of y for the next BN. */
int raae_sum3_fwd(const raae_view_t* a, const raae_view_t* b, const raae_view_t* c, int B, int C, int L, float* y,
                  double* out_partials, int* out_nparts, void* stream);
/* dRaw of a grad spec written (accumulate = 0) or added (accumulate != 0) to `draw` -- the identity
 * shortcut of a block without conv_short (model.py:83), and the gradient handed to a dense layer;
 * dslope (may be NULL): PReLU-slope gradient [C] as *nslab slabs. */
int raae_grad_materialize(const raae_grad_t* go, int B, int C, int L, float* draw, int accumulate, float* dslope,
                          long slab_stride, int* nslab, void* stream);

/* ---- fused residual-block forward (EncodingBlock / DecodingBlock, reference model.py:24-174) ----
 * BatchNorm (batch statistics) is the only grid-wide dependency inside a block, so its forward is two kernels:
 *   A: R = bn1(X) staged once per sample -> T1 = conv1(R) (+stats of PReLU1), Sh = conv_short(R),
 *      E1 = fc1(dropout(R)), E2 = fc2(PReLU(E1)) (+stats of PReLU when pE2 != NULL)
 *   B: T2 = conv2(bn2(PReLU1(T1))), E3 = conv_excit(bn_e(PReLU(E2))), Y = PReLU2(T2) + PReLUs(Sh)|R + PReLUe(E3|E2)
 *      (+stats of Y).  Trailing "reserved" fields are filled by the library.  Needs Cin, Cout <= 8. */
typedef struct {
    raae_view_t in;            /* block input X through bn1 (has_bn = 0 if the block has none); in.mask = NULL */
    const float* mask;         /* dropout scale of the excitation branch [B][Cin][Lin] or NULL */
    int B;
    int Cin, Cout, Lin, L1, Lout, E;
    raae_conv_t cv1, cvs; int has_short;
    const float *w1, *b1, *slope1, *ws, *bs, *wf1, *bf1, *se1, *wf2, *bf2, *se2;
    float *T1, *Sh, *E1, *E2;
    double *pT1, *pE2;
    int S, ngroups, sh_lin, sh_l1, sh_lout, sh_e, halo;      /* reserved */
} raae_block_fwd_a_t;
typedef struct {
    raae_view_t vT1, vE2, vR;  /* T1 via PReLU1+bn2; E2 via PReLU_e2 (+bn_excit); X via bn1 (identity shortcut only) */
    int B, Cin, Cout, L1, Lout;
    raae_conv_t cv2, cve; int has_short, has_excit;
    const float *w2, *b2, *slope2, *we, *be, *se3, *Sh, *ss;
    float *T2, *E3, *Y; double* pY;
    int S, ngroups, sh_l1, sh_lout, halo2;                   /* reserved */
} raae_block_fwd_b_t;
/* *nparts receives the number of partial-sum rows written to pT1 / pE2 (A) or pY (B). */
int raae_block_fwd_a(const raae_block_fwd_a_t* a, int* nparts, void* stream);
int raae_block_fwd_b(const raae_block_fwd_b_t* a, int* nparts, void* stream);

/* raae_block_fwd_a / raae_block_fwd_b of TWO residual blocks that do not depend on each other, in ONE launch: of the
 * six encoder and four decoder forwards of a step (trainer.py:113-200) two are run only for their BatchNorm /
 * RNG side effects, and each of those chains runs beside a chain that is needed.  Same arguments and outputs
 * as the two single calls. */
int raae_block_fwd_a2(const raae_block_fwd_a_t* x, const raae_block_fwd_a_t* y, int* nparts_x, int* nparts_y,
                      void* stream);
int raae_block_fwd_b2(const raae_block_fwd_b_t* x, const raae_block_fwd_b_t* y, int* nparts_x, int* nparts_y,
                      void* stream);

/* ---- fused residual-block backward, data gradients (mirror of the forward split) ----
 *   B: dY = BNbwd(gy) -> dT2, dSh (= dY when the shortcut is the identity), dEx (= dE3, or dE2 when the block has no
 *      conv_excit) materialised once; dBn2 = conv2^T dT2 and dBnE = conv_excit^T dE3 with their BatchNorm-backward
 *      partial sums; PReLU-slope gradient slabs of relu2, relu_short, relu_excit_3 (or relu_excit_2).
 *   A: dT1, dE2 (when conv_excit exists), dE1 materialised; dR = conv1^T dT1 + short^T dSh | dSh + mask * fc1^T dE1
 *      (+ partial sums for bn1; dR == NULL skips it); slope slabs of relu1, relu_excit_2, relu_excit_1.
 * The conv / fc WEIGHT gradients are then produced by raae_conv_bwd_weight / raae_lenlin_bwd_weight from the
 * materialised gradients (no BatchNorm prologue).  Slab s of a slope tensor is at ptr + s*slab_stride. */
typedef struct {
    raae_grad_t gy;            /* gradient arriving at the block output Y: g [+ g_partials, bn = stats of Y, u = Y] */
    raae_view_t vT1, vE2;      /* T1 via PReLU1+bn2, E2 via PReLU_e2+bn_excit (for the BN-backward sums) */
    int B, Cin, Cout, L1, Lout;
    raae_conv_t cv2, cve; int has_short, has_excit;
    const float *w2, *we, *slope2, *ss, *se, *T2, *Sh, *Ex;
    float *dT2, *dSh, *dEx, *dBn2, *dBnE;
    double *pdBn2, *pdBnE;
    float *dslope2, *dslope_s, *dslope_e;
    long slab_stride;
    int S, ngroups, sh_l1, sh_lout;                          /* reserved */
} raae_block_bwd_b_t;
typedef struct {
    raae_grad_t g1, ge;        /* dBn2 through bn2/PReLU1 (raw = T1); dBnE through bn_excit/PReLU_e2 (raw = E2) */
    raae_view_t in;            /* block input X through bn1; in.mask = NULL */
    const float* mask;
    int B, Cin, Cout, Lin, L1, Lout, E;
    raae_conv_t cv1, cvs; int has_short, has_excit;
    const float *w1, *ws, *wf1, *wf2, *se1, *E1, *dSh;
    float *dT1, *dE2, *dE1, *dR;
    double* pdR;
    float *dslope1, *dslope_e2, *dslope_e1;
    long slab_stride;
    int S, ngroups, sh_lin, sh_l1, sh_lout, sh_e;            /* reserved */
} raae_block_bwd_a_t;
/* *nparts: partial-sum rows AND slope slabs written (= workgroups launched). */
int raae_block_bwd_b(const raae_block_bwd_b_t* a, int* nparts, void* stream);
int raae_block_bwd_a(const raae_block_bwd_a_t* a, int* nparts, void* stream);

/* All conv / fc WEIGHT gradients of one residual block in ONE launch (up to 4 conv + 2 length-linear tasks run
 * side by side on disjoint workgroup ranges).  Gradients must be direct (materialised by raae_block_bwd_a/b):
 * go.has_bn = 0, go.slope = NULL.  nslab[i] receives the slab count of task i (conv tasks first, then lin). */
typedef struct { raae_grad_t go; raae_conv_t cv; raae_view_t in; float* dw; float* dbias; } raae_wgrad_conv_t;
typedef struct { raae_grad_t go; int C, E, Lin; raae_view_t in; float* dw; float* dbias; } raae_wgrad_lin_t;
typedef struct {
    int n_conv, n_lin, B;
    long slab_stride;
    raae_wgrad_conv_t conv[4];
    raae_wgrad_lin_t lin[2];
} raae_block_wgrad_t;
int raae_block_wgrad(const raae_block_wgrad_t* a, int* nslab, void* stream);

/* raae_block_bwd_b of one residual block and raae_block_wgrad of the block AFTER it (model.py:24-174 + autograd;
 * its data gradients are complete by then) in ONE launch: the two are independent and overlap on the chip, which
 * at small batches is cheaper than parallel graph branches.  Same arguments and outputs as the two calls. */
int raae_block_bwd_b_wgrad(const raae_block_bwd_b_t* b, const raae_block_wgrad_t* w, int* nparts, int* nslab,
                           void* stream);

/* Data parallel (replaces the reference's ipyparallel trial farm, sc/cmd/train_sc.py:25-45, per the
 * north star): out[i] = fixed-order sum of the slabs of element i -- the flat gradient that is then
 * averaged across ranks with one RCCL all-reduce per phase and fed to raae_adam_step as a single slab. */
int raae_slab_reduce(const float* g_slabs, long slab_stride, const unsigned short* seg_nslab, long n, float* out,
                     int max_nslab, void* stream);
/* once per training step: steps[i] += 1 for every bit i set in mask; rng_counter[0] += 1;
 * cursor[0] += cursor_inc (epoch row cursor of raae_gather_batch).  rng_counter points at the engine's
 * {counter, seed, keys} words (three 8-byte words; may be NULL): the step's dropout hash keys (raae_maskgen_t.keys)
 * are stored at rng_counter + 2. */
int raae_step_tick(int* steps, int n, unsigned mask, unsigned long long* rng_counter, int* cursor, int cursor_inc,
                   void* stream);

/* Fill of the per-step random tape (speed mode; parity mode uploads a tape drawn on the host in the reference's order,
 * SURVEY.md 3.4).  seg_desc (device): [nseg][4] ints {offset, count, kind, hash offset}, offsets multiples of 4,
 * ascending; kind 0 = N(0,1) (Philox4x32-10 + Box-Muller), 1 = dropout scale {0, 1/keep}, 2 = keep flags {0, 1} as
 * bf16 (two per float); kinds 1 and 2 evaluate the counter-based hash of raae_maskgen_t at hash offset + element
 * index, so a slot has the same values on the tape as when its consumer generates it; seg_scale (device) [nseg] =
 * keep probability; counter (device) = step index. */
int raae_rng_fill(float* tape, const int* seg_desc, const float* seg_scale, int nseg, long total,
                  unsigned long long seed, const unsigned long long* counter, void* stream);

/* The head of a training step in one launch: raae_step_tick + raae_rng_fill + raae_gather_batch (trainer.py:106-113: the
 * DataLoader's next batch, ``spec_in += randn_like(spec_in) * spec_noise``).  rng_state (device) = {step counter, seed, hash keys};
 * the launch works with counter + 1 and cursor + stride and its last workgroup stores them (and steps[i] += 1 for the
 * bits of step_mask) for the kernels that follow.  noise_tape: the noise slot of a host-filled tape (parity mode) or
 * NULL: N(0, 1) generated in the kernel at position noise_goff of the step's Gaussian numbering -- bit for bit what
 * raae_rng_fill writes for that slot.  nseg > 0: the tape's resident slots are filled as by raae_rng_fill.
 * ticket: device unsigned, zero before the first call. */
typedef struct {
    int* steps; int nsteps; unsigned step_mask; unsigned long long* rng_state; int* cursor; int stride; unsigned* ticket;
    const float* spec; const float* aux; const long* idx; int B, L, n_aux; float spec_noise;
    const float* noise_tape; long noise_goff; float* spec_out; float* aux_out;
    float* tape; const int* seg_desc; const float* seg_scale; int nseg; long total;
} raae_step_begin_t;
int raae_step_begin(const raae_step_begin_t* p, void* stream);

/* ---- independent trials batched into one launch (reference: sc/cmd/train_sc.py:127-143 maps `trials` over
 * engines) ----
 * The entry points of the dense-network path (raae_step_begin, raae_dense_fwd_s / _fwd2 / _bwd_s, raae_style_bn_*,
 * raae_disc_fused, raae_rank_loss_fwd_bwd, the three loss kernels, raae_adam_step) exist in a second form whose grid
 * plane z works on trial z's argument block.  raae_record_begin/end log the launches one trial makes on the calling
 * thread (they still run); raae_multi_build takes the logs of T structurally identical trials and uploads, launch by
 * launch, the T argument blocks as one table (RAAE_EINVAL when the logs differ in kernel instance, geometry or LDS);
 * raae_multi_launch replays the program with gridDim.z = T on `stream` (capturable).  A trial's arithmetic is the body
 * it runs alone: results are bit-identical. */
int raae_record_begin(void);
int raae_record_end(void** handle, int* n_launches);
int raae_record_free(void* handle);
int raae_multi_build(void* const* handles, int T, void** program);
int raae_multi_launch(void* program, void* stream);
int raae_multi_count(void* program);
int raae_multi_free(void* program);

/* ---- launch-geometry hint (round 3; thread-local) ----------------------------------------------------------------------
 * The conv-network entry points size a workgroup's group of samples from the batch they are called with: at 256 rows a
 * launch is 64-256 small workgroups, right for one trial alone.  A caller that launches the kernels of T trials at once
 * (raae_multi_*) wants each trial's launch to look like its share of a T-times larger batch -- fewer, longer-running
 * workgroups, fewer partial-statistic rows for every consumer: raae_tile_hint(m) makes the calling thread's following
 * launches size their groups as if the batch were m times larger (1 <= m <= 64; 1 = default).  It changes the
 * summation order of batch reductions (never which instance runs), so the SAME hint must be used wherever bitwise
 * equality is expected.  A block's weight-gradient tasks get 128 / m workgroups (= gradient slabs) each.  Measured, conv
 * networks, 256 rows: m = 4: 8 batched trials 2245 -> 2995 aggregate steps/s, 16 trials 2478 -> 3667; one trial alone
 * 782 -> 620. */
int raae_tile_hint(int rows_multiplier);

/* ---- hand-over of a step's deferred tail (round 3; rankaae_amd.engine.StepEngine `overlap_steps`) ---------------------
 * The smoothness phase of step k (trainer.py:189-200) updates only the decoder; after its encoder forward nothing in it
 * reads encoder state, and nothing in phase A of step k+1 (trainer.py:113-127: encoder forward, discriminator, encoder
 * backward) reads decoder state.  The engine therefore runs that tail beside the next step's phase A, as a second
 * branch of the next step's graph.  raae_tail_prepare ends the step's head: dst[0..n) = src[0..n) (the styles, whose
 * buffer the next forward overwrites), *step_counter += 1 (the tail optimizer's step count; the step head then leaves
 * that bit out of its mask), and the two 32-bit dropout hash keys at rng_state + 2 copied to tail_state + 2 (the
 * tail's dense kernels take their raae_maskgen_t.keys there).  Any pointer but src / dst may be NULL. */
int raae_tail_prepare(const float* src, float* dst, long n, int* step_counter, const unsigned long long* rng_state,
                      unsigned long long* tail_state, void* stream);

/* ---- stream / graph / event plumbing (HIP runtime; used by the engine and bench.py) ---- */
int raae_graph_begin(void* stream);
int raae_graph_end(void* stream, void** graph_exec);
int raae_graph_launch(void* graph_exec, void* stream);
int raae_graph_destroy(void* graph_exec);
int raae_event_create(void** ev);
int raae_event_record(void* ev, void* stream);
int raae_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
int raae_event_destroy(void* ev);
int raae_stream_sync(void* stream);
const char* raae_error_string(int code);
int raae_device_info(int* cu_count, int* lds_bytes, char* name, int name_len);
#define RAAE_ABI_VERSION 16
int raae_abi_version(void);
/* First 16 hex digits of sha256 over include/rankaae_hip.h + csrc/raae_*.{h,inc,hip} at build time
 * (build.sh); the Python loader recomputes it and refuses a library built from other sources. */
const char* raae_source_digest(void);

#ifdef __cplusplus
}
#endif
#endif
