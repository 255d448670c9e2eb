/*
 * deepsolid_hip.h -- C ABI of libdeepsolid_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the VMC inner loop of bytedance/DeepSolid.  The reference
 * has no FFI layer: the boundary is three Python factories whose closures are
 * consumed by process.py / train.py.  Each entry point below names the reference
 * function(s) whose arithmetic it replaces; the Python side
 * (deepsolid_amd/{network,hamiltonian,qmc,ewaldsum,distance}.py) keeps the
 * reference signatures and calls these through ctypes with raw device pointers.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; ds_last_error()
 *     returns a thread-local message for the last failure;
 *   - all `const void*` / `void*` data arguments are DEVICE pointers owned by the
 *     caller (torch tensors), element type given by ds_system_desc.dtype
 *     (0 = float64, 1 = float32); walkers are (B, 3*N) row-major, B = batch;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); nothing is
 *     allocated or synchronised after ds_system_create.  A handle carries device
 *     scratch of its own (the int8 digit planes of the hidden-layer weights, refilled
 *     by the first launch of every call that takes `params`): calls on ONE handle
 *     must be issued from one host thread and complete in issue order (one stream, or
 *     streams ordered by events); for concurrent streams create one handle per stream;
 *   - no torch types appear anywhere in this interface.
 */
#ifndef DEEPSOLID_HIP_H
#define DEEPSOLID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_MAX_LAYERS 8
#define DS_MAX_SYM 6 /* rows of AV/BV: 3 ('minimal'), 4 (fcc/hexagonal), 6 (bcc) -- supercell.py:98-140 */

typedef struct ds_system ds_system; /* opaque handle, one per device */

/* Host-side description of one simulation cell + network architecture.  All
 * pointers are HOST pointers to float64 / int data and are copied by
 * ds_system_create.  Field provenance (reference file:line):
 *   n_up,n_dn            simulation_cell.nelec                      network.py:644
 *   prim_* / sim_*       simulation_cell.original_cell.a/.AV/.BV,
 *                        simulation_cell.a/.AV/.BV                  network.py:281-296
 *   prim_atoms           original_cell.atom_coords()                network.py:643
 *   klist_up/dn          hf.SCF.klist                               hf.py:84-104
 *   hidden_single/double, n_det, ...  cfg.network.detnet            base_config.py:129-139
 *   ewald_*              EwaldSum.__init__ products                 ewaldsum.py:34-136
 *   dist_mode            MinimalImageDistance dispatch              distance.py:43-61
 */
typedef struct ds_system_desc {
    int32_t dtype;               /* 0 = f64, 1 = f32 */
    int32_t n_up, n_dn;
    int32_t n_atoms_prim;
    const double* prim_atoms;    /* (n_atoms_prim, 3) */
    double prim_a[9];
    double sim_a[9];
    int32_t n_sym;               /* rows of AV/BV */
    double prim_AV[DS_MAX_SYM * 3], prim_BV[DS_MAX_SYM * 3];
    double sim_AV[DS_MAX_SYM * 3], sim_BV[DS_MAX_SYM * 3];
    /* network */
    int32_t n_layers;            /* len(hidden_dims) */
    int32_t hidden_single[DS_MAX_LAYERS];   /* the REFERENCE's hidden_dims[l][0] verbatim (network.py:111-132): 1..1024 */
    int32_t hidden_double[DS_MAX_LAYERS];   /* ... hidden_dims[l][1]: 1..32.  The library pads both internally (ds_device_widths) */
    int32_t n_det;               /* determinants */
    int32_t distance_type;       /* 0 = 'nu' (network.py:207), 1 = 'tri' (network.py:227; isotropic envelope only) */
    int32_t envelope_type;       /* 0 = 'isotropic' (network.py:335), 1 = 'diagonal' (:340), 2 = 'full' (:358) */
    int32_t full_det;            /* 0 block-diagonal determinants per spin (base_config default), 1 dense N x N (network.py:558) */
    int32_t use_last_layer;      /* 1: the orbital head takes the symmetric features of the last layer (network.py:535) */
    int32_t bias_orbitals;       /* 1: orbital[s]['b'] is added to the orbital head (network.py:181-184) */
    const double* klist_up;      /* (n_up, 3) */
    const double* klist_dn;      /* (n_dn, 3) */
    /* Ewald tables */
    int32_t n_atoms_sim;
    const double* sim_atoms;     /* (n_atoms_sim, 3) */
    const double* sim_charges;   /* (n_atoms_sim,) */
    int32_t dist_mode;           /* 0 diagonal, 1 orthogonal, 2 general */
    int32_t n_g;
    const double* gpoints;       /* (n_g, 3) */
    const double* gweight;       /* (n_g,) */
    const double* ion_exp_re;    /* (n_g,)  Re sum_a Z_a exp(i G.R_a) */
    const double* ion_exp_im;    /* (n_g,) */
    double ewald_alpha;
    double ee_const;             /* EwaldSum.ee_const(N)   ewaldsum.py:109 */
    double ei_const;             /* EwaldSum.ei_const(N)   ewaldsum.py:112 */
    double ii_total;             /* ion_ion + ii_const     ewaldsum.py:190 */
} ds_system_desc;

/* Parameter buffer: one flat device array of `dtype` elements.  ds_param_layout
 * reports, for the handle's architecture, the element offset and the logical
 * (rows, cols) of every block in this order:
 *   for l in layers:  W_loc_l  [(h1_in + nch*h2_in), h1_out]   rows: h_i | mean_j h2_ij (up) | (dn)
 *                     W_sh_l   [(nch*h1_in), h1_out]           rows: mean_up h | mean_dn h
 *                     b_l      [1, h1_out]
 *   for l in double layers: W2_l [h2_in, h2_out], b2_l [1, h2_out]        (n_layers-1 of them, n_layers with use_last_layer)
 *   for s in active spins:  W_orb_s [h1_last (+ nch*h2_last with use_last_layer), cols]  cols = 2*norb*n_det rounded up to 64,
 *                             norb = n_s (N with full_det); Re/Im columns interleaved per 16-column tile, see
 *                             deepsolid_amd/device.py::_orbital_column_map and csrc/ds_gemm.h::orb_col
 *                           [W_sh_orb_s [nch*h1_last, cols]   only with use_last_layer]
 *                           [b_orb_s [1, 2*norb*n_det]        only with bias_orbitals]
 *                           pi_s [A, norb*n_det], sigma_s [A | 3A | 9A, norb*n_det] (isotropic | diagonal | full)
 * Layer-0 rows are padded with zero rows to multiples of 4 (only 'tri', 7 features, needs it).
 * i.e. the rows of the reference's `single[l]['w']` (network.py:126-158) split
 * into the per-electron and the shared (spin-mean) part. */
typedef struct ds_param_block {
    int64_t offset;
    int32_t rows, cols;
} ds_param_block;

/* The widths the kernels run for the reference's hidden_dims, and where a residual is added (host code, needs no GPU).
 * One-electron widths are zero-padded to multiples of 64, pair widths to 16 or 32 -- exact: a padded feature is tanh(0) = 0 in
 * every layer and meets zero rows in the next.  res_single[l] / res_double[l] = 1 where the reference adds a residual
 * (network.py:525-528: input width == output width of the REFERENCE's widths), independent of the padded widths.
 * n_in_single / n_in_double: widths of the input features (nf * atoms, nf; nf = 4 for 'nu', 7 for 'tri'); n_double: pair layers
 * that run (n_layers - 1, or n_layers with use_last_layer).  ds_system_create applies exactly this to its descriptor;
 * ds_param_layout reports the PADDED block shapes: a caller copies each reference block into the top-left corner of its
 * padded segments ([h | pair-up | pair-dn] rows of W_loc, [mean-up | mean-dn] rows of W_sh, ...) and leaves the rest zero. */
int ds_device_widths(const int32_t* hidden_single, const int32_t* hidden_double, int32_t n_layers, int32_t n_in_single,
                     int32_t n_in_double, int32_t n_double, int32_t* dev_single, int32_t* dev_double, int32_t* res_single,
                     int32_t* res_double);

int ds_system_create(const ds_system_desc* desc, ds_system** out);
void ds_system_destroy(ds_system* sys);
const char* ds_last_error(void);

/* Number of dense hidden one-electron layers per local-energy evaluation whose per-electron contraction runs as a 47-bit
 * truncating fixed-point split on the int8 matrix pipe (csrc/ds_i8.h: float64 in, float64 out; 5-slot-tile float64 cells with 256
 * features).  Opt-in since round 6 (environment DS_I8=1 at ds_system_create): 0 otherwise, or when the architecture has no instance. */
int ds_int8_layers(const ds_system* sys);

int64_t ds_param_count(const ds_system* sys);
/* fills up to `max_blocks` entries, returns the number of blocks */
int ds_param_layout(const ds_system* sys, ds_param_block* blocks, int max_blocks);

/* bytes of scratch the calls below need for a batch of B walkers */
int64_t ds_workspace_bytes(const ds_system* sys, int64_t B);

/* network.eval_func (network.py:563-606), methods eval_phase_and_slogdet /
 * eval_slogdet / eval_logdet, batched as process.py:116-118 vmaps it.
 * out_logabs (B,), out_phase (B,2) = (Re, Im) of the unit phase; either may be NULL. */
int ds_logpsi(ds_system* sys, const void* params, const void* x, int64_t B,
              void* out_logabs, void* out_phase, void* ws, int64_t ws_bytes, void* stream);

/* value and gradient of log psi w.r.t. the walker coordinates: what qmc.make_mcmc_step builds with
 * jax.vmap(jax.value_and_grad(importance_sampling, argnums=1)) (qmc.py:324) for importance_update
 * (qmc.py:83-150).  out_grad (B, 3N, 2): Re = grad log|psi|, Im = grad arg psi.  Uses the
 * forward-Laplacian chain (the gradient is its trace stage); out_logabs / out_phase optional. */
int ds_logpsi_grad(ds_system* sys, const void* params, const void* x, int64_t B,
                   void* out_logabs, void* out_phase, void* out_grad, void* ws, int64_t ws_bytes, void* stream);

/* Parameter gradient of  L = sum_b [ cot[b][0] * log|psi_b| + cot[b][1] * arg psi_b ]  (reverse sweep over
 * the value chain).  This is the vector-Jacobian product the reference obtains from jax.jvp / jax.grad of
 * batch_network inside the custom JVP of total_energy (train.py:91-142): with cot = clip_diff / B (Re, Im)
 * the result is  tangents_dot = mean(Re(clip_diff * conj(d log psi)))  for every parameter direction, i.e.
 * the energy gradient jax.value_and_grad(total_energy) returns (train.py:147-176, process.py:226-228).
 * cot (B, 2); grad (ds_param_count(),) in the packed layout of ds_param_layout, overwritten (padding entries
 * are zero); out_logabs / out_phase optional.  Every network option of ds_system_desc is supported.
 * Weight gradients are reduced over
 * walkers in a fixed order (no atomics): the result is bit-reproducible run to run. */
int64_t ds_vjp_workspace_bytes(const ds_system* sys, int64_t B);
int ds_logpsi_vjp(ds_system* sys, const void* params, const void* x, int64_t B, const void* cot, void* grad,
                  void* out_logabs, void* out_phase, void* ws, int64_t ws_bytes, void* stream);

/* network.eval_func method 'eval_mats' (network.py:601): out_up (B, n_det, n_up, n_up, 2),
 * out_dn (B, n_det, n_dn, n_dn, 2), complex as (Re, Im) pairs. */
int ds_orbitals(ds_system* sys, const void* params, const void* x, int64_t B,
                void* out_up, void* out_dn, void* ws, int64_t ws_bytes, void* stream);

/* ewaldsum.EwaldSum.energy (ewaldsum.py:185-191): out (B,3) = (ee, ei, ii). */
int ds_ewald(ds_system* sys, const void* x, int64_t B, void* out, void* stream);

/* hamiltonian.local_energy_seperate (hamiltonian.py:194-228), all Laplacian modes
 * (for / dim_batch / hessian / partition return the same numbers and map to one
 * forward-Laplacian kernel chain).  out_ke (B,2) = complex kinetic energy,
 * out_ewald (B,) = ee+ei+ii; out_logabs (B,) / out_phase (B,2) optional (NULL). */
int ds_local_energy(ds_system* sys, const void* params, const void* x, int64_t B,
                    void* out_ke, void* out_ewald, void* out_logabs, void* out_phase,
                    void* ws, int64_t ws_bytes, void* stream);

/* distance.enforce_pbc (distance.py:144-163), batched over B*N electrons; needs no handle.
 * latvec: HOST pointer to the 3x3 lattice (rows = lattice vectors); dtype 0 = f64, 1 = f32;
 * x, out_x: (n_elec, 3) device arrays; out_wrap (n_elec, 3) device array or NULL. */
int ds_enforce_pbc(const double* latvec, int dtype, const void* x, int64_t n_elec, void* out_x,
                   void* out_wrap, void* stream);

/* qmc.mh_update symmetric branch (qmc.py:190-196, 217-222) split around the network call:
 *   propose: x2 = wrap(x1 + width * normal)
 *   accept : cond = (lp2 - lp1) > log(uniform); x1,lp1 <- select; n_accept[0] += sum(cond)
 * `normal` (B,3N) and `uniform` (B,) are caller-supplied noise (torch Philox on device). */
int ds_mh_propose(ds_system* sys, const void* x1, const void* normal, double width, int64_t B,
                  void* x2, void* stream);
int ds_mh_accept(ds_system* sys, void* x1, void* lp1, const void* x2, const void* lp2,
                 const void* uniform, int64_t B, void* n_accept, void* stream);

/* The reference's other proposals around the network call, per move, with caller-supplied noise (both are flagged "untested" in
 * base_config.py:122-126):
 *   mode 1: asymmetric all-electron move of qmc.mh_update with `atoms` (qmc.py:197-215): x2 = wrap(x1 + width * hmean(x1) * normal),
 *           hmean = harmonic mean of an electron's non-periodic distances to the nuclei aux (n_aux, 3) (qmc.py:44-60); the test adds
 *           the reverse / forward proposal densities (_log_prob_gaussian, qmc.py:26-42);
 *   mode 2: drift-biased move of qmc.importance_update (qmc.py:83-124): x2 = wrap(x1 + width * normal + width^2 * limdrift(grad)),
 *           aux / aux1 = grad log|psi| at x1 (B,3N), aux2 = the same at x2 (both from ds_logpsi_grad); lp <- 2 log|psi(x2)| +
 *           (|gauss|^2 - |gauss + width^2 (limdrift g1 + limdrift g2)|^2) / (2 width^2).
 *           `scratch`: 2 device elements shared by the two calls of one move -- the batch maxima of |grad| that limdrift's clip
 *           uses as its upper bound (qmc.py:78; it binds only when every drift of the batch is below the cutoff).
 * ds_mh_accept_ex takes log|psi(x2)| (B,), selects x1 / lp1 in place and increments n_accept. */
int ds_mh_propose_ex(ds_system* sys, int mode, const void* x1, const void* normal, double width, const void* aux, int n_aux,
                     int64_t B, void* x2, void* scratch, void* stream);
int ds_mh_accept_ex(ds_system* sys, int mode, void* x1, void* lp1, const void* x2, const void* logabs2, const void* uniform,
                    const void* normal, double width, const void* aux1, const void* aux2, int n_aux, int64_t B,
                    void* n_accept, void* scratch, void* stream);

/* qmc.make_mcmc_step's jitted loop (qmc.py:335-362) for the default sampler (all-electron symmetric mh_update,
 * qmc.py:153-196,217-222): `steps` moves enqueued back to back on `stream`, no host synchronisation:
 *     [lp = 2 log|psi(x)|  if !lp_valid (:357)]
 *     repeat steps times:  x2 = wrap(x + width * N(0,1));  lp2 = 2 log|psi(x2)|;
 *                          accept iff lp2 - lp > log u;  x, lp <- select;  n_accept[0] += #accepted
 * Noise: a counter-based Philox4x32-10 generator evaluated inside the kernels replaces jax.random.split / normal /
 * uniform (:190-192, :217-218): the draw for (move i, electron e | walker w) is a pure function of
 * (philox_seed, philox_offset + i, index), so runs are reproducible and ranks decorrelate by seed.  A caller
 * advances philox_offset by `steps` between calls.  Test mode: `normals` (steps, B, 3N) and `uniforms` (steps, B)
 * device arrays replay caller-supplied noise instead (both or neither).
 * x (B,3N) and lp (B,) are updated in place; n_accept (1,) of dtype is incremented (pmove = n_accept / (steps * B),
 * qmc.py:360).  Workspace: ds_mcmc_workspace_bytes(sys, B). */
int64_t ds_mcmc_workspace_bytes(const ds_system* sys, int64_t B);
int ds_mcmc_step(ds_system* sys, const void* params, void* x, void* lp, int64_t B, int steps, double width,
                 uint64_t philox_seed, uint64_t philox_offset, const void* normals, const void* uniforms,
                 int lp_valid, void* n_accept, void* ws, int64_t ws_bytes, void* stream);
/* The same loop with ONE-ELECTRON moves -- qmc.mh_one_electron_update (qmc.py:227-287) as make_mcmc_step drives it
 * (qmc.py:355-358: nsteps = N * steps, move i displaces electron i % N): move i of this call displaces electron
 * (first_electron + i) % N of every walker, wraps the whole configuration, evaluates log|psi| and selects.
 * Philox index of the displaced electron's normals = w * N + electron (the all-electron stream's index of that
 * electron); test mode: `normals` (moves, B, 3), `uniforms` (moves, B).  pmove = n_accept / (moves * B). */
int ds_mcmc_step_one_electron(ds_system* sys, const void* params, void* x, void* lp, int64_t B, int moves, int first_electron,
                              double width, uint64_t philox_seed, uint64_t philox_offset, const void* normals,
                              const void* uniforms, int lp_valid, void* n_accept, void* ws, int64_t ws_bytes, void* stream);
/* The same loop with the drift-biased importance-sampled move -- qmc.importance_update (qmc.py:83-150, symmetric branch) as
 * make_mcmc_step(importance_sampling=...) drives it: per move  g1 = grad log|psi|(x);  x2 = wrap(x + width * N + width^2 *
 * limdrift(g1));  (log|psi|, g2) at x2;  accept with the forward / reverse proposal densities (qmc.py:119-137).
 * Noise as in ds_mcmc_step: Philox (normals of electron e: index w * N + e, uniform of walker w), or explicit
 * `normals` (steps, B, 3N) / `uniforms` (steps, B).  Workspace: ds_mcmc_workspace_bytes(sys, B). */
int ds_mcmc_step_importance(ds_system* sys, const void* params, void* x, void* lp, int64_t B, int steps, double width,
                            uint64_t philox_seed, uint64_t philox_offset, const void* normals, const void* uniforms,
                            int lp_valid, void* n_accept, void* ws, int64_t ws_bytes, void* stream);
/* The same loop with the asymmetric proposal of mh_update(atoms=...) (qmc.py:197-215): the step width of an electron is
 * width x the harmonic mean of its (non-periodic) distances to `atoms` (n_atoms, 3) [device, dtype]; the forward / reverse
 * proposal densities enter the acceptance.  Noise as in ds_mcmc_step.  Workspace: ds_mcmc_workspace_bytes(sys, B). */
int ds_mcmc_step_asymmetric(ds_system* sys, const void* params, void* x, void* lp, int64_t B, int steps, double width,
                            const void* atoms, int n_atoms, uint64_t philox_seed, uint64_t philox_offset, const void* normals,
                            const void* uniforms, int lp_valid, void* n_accept, void* ws, int64_t ws_bytes, void* stream);
/* the raw Philox block ds_mcmc_step uses for (seed, offset + step, index, stream_id): host evaluation for tests
 * (stream_id 0 / 1: normal deviates of electron `index`, 2: the uniform deviate of walker `index`). */
void ds_philox_host(uint64_t seed, uint64_t offset, uint64_t step, uint64_t index, int stream_id, uint32_t out[4]);

/* Packed batch statistics of train.make_loss.total_energy (train.py:74-82) in one deterministic reduction:
 * out_stats (8,) float64 on the device =
 *   [ sum Re E_L, sum Im E_L, sum |E_L|^2, n, n_nonfinite, sum Re E_kin, sum Im E_kin, sum E_ewald ],  E_L = ke + ewald.
 * The vector is what crosses GPUs in ONE all-reduce (replacing the pmeans of train.py:78-80); n_nonfinite is the
 * count behind the reference's optional check_nan step rejection (process.py:303-318). */
int ds_energy_stats(ds_system* sys, const void* ke, const void* ewald, int64_t B, double* out_stats, void* stream);

/* Stage dumps for parity tests (tests only; sizes documented in DESIGN.md).
 * Runs the forward-Laplacian chain for the first walkers of the batch and copies
 * the named intermediate into `out` (device, dtype elements).  Returns elements written. */
int64_t ds_debug_stage(ds_system* sys, const void* params, const void* x, int64_t B,
                       const char* stage, void* out, int64_t out_elems,
                       void* ws, int64_t ws_bytes, void* stream);

/* Per-kernel timing with HIP events recorded on the caller's stream around every launch of the
 * local-energy chain (bench.py's roofline numbers).  ds_profile_enable(sys, 1) resets and starts for every
 * kernel, ds_profile_enable(sys, 2 + kind) for one kernel kind only (no events around the others), 0 stops;
 * ds_profile_read synchronises the recorded events and returns, per kernel kind, the summed
 * duration in ms and the number of launches (arrays of DS_PROF_KINDS entries). */
#define DS_PROF_FEATURES 0
#define DS_PROF_M2_EXPAND 1
#define DS_PROF_TWO_LAYER 2
#define DS_PROF_SINGLE_FIRST 3   /* one-electron layer 0 (K = 4A + 8): k_layer0_stats / k_jet_gemm<.,9> + k_layer0_means in front of the low-rank layer 1, else k_jet_gemm<.,1> */
#define DS_PROF_SINGLE_HIDDEN 4  /* k_jet_gemm<.,2> of the dense hidden one-electron layers (K = 256 + 64): the dominant kernel */
#define DS_PROF_ORBITAL 5        /* k_jet_gemm<.,5> of the orbital head (fused envelope x phase epilogue) */
#define DS_PROF_DET_INVERSE 6
#define DS_PROF_DET_TRACE 7
#define DS_PROF_COMBINE 8
#define DS_PROF_EWALD 9
#define DS_PROF_SHARED_TERM 10    /* per-walker spin-mean term S = W_sh^T mean_i h_i (k_shared_term; layer 0: k_jet_gemm<.,0>) */
#define DS_PROF_SINGLE_LR 11     /* k_layer1_lr: the first hidden layer on the low-rank form of the layer-0 output */
#define DS_PROF_KINDS 12
int ds_profile_enable(ds_system* sys, int on);
int ds_profile_read(ds_system* sys, double* ms_total, int64_t* launches);
/* In-kernel clock probe of the dominant kernel (hidden one-electron layers), active while profiling is enabled for it: one
 * wave per workgroup reads the shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) at entry
 * and exit; the sums over all workgroups since ds_profile_enable are returned.  shader_cycles / ref_ticks x 100 MHz is the
 * clock the kernel actually ran at in that region (the MFMA peak scales with it). */
int ds_profile_read_clock(ds_system* sys, double* shader_cycles, double* ref_ticks);
/* Kernel-development aid: shader-clock stamps written by one wave of the hidden-layer kernel at its phase boundaries when the
 * library runs with DS_DBG=32 and profiling is enabled (tools/gemm_timeline.py); n <= 1022 stamps. */
int ds_debug_timeline(ds_system* sys, uint64_t* out, int n);

/* Calibration kernel for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters: copies n_elems float64
 * with the 8-byte-per-lane access pattern of the jet tensors (known traffic = 8 n read + 8 n written). */
int ds_calib_copy(const void* src, void* dst, int64_t n_elems, void* stream);

/* fp64 MFMA issue-rate micro-benchmark used to confirm the roofline peak: every wave issues
 * `iters` x `n_acc` independent v_mfma_f64_16x16x4_f64 (n_acc = 1,2,4,8,16 accumulators), with
 * `blocks_per_cu` waves resident per SIMD; returns the FLOPs executed (caller times with HIP events). */
int64_t ds_mfma_f64_peak(int64_t iters, int blocks_per_cu, int n_acc, void* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPSOLID_HIP_H */
