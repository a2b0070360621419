// The reference's own UIPC scene on the GPU (second slice; included by fem_kernels.hip inside namespace tacex):
// the gelpad (Stable Neo-Hookean tets, soft position constraints) + ONE FREE AFFINE-BODY BALL per env + the ground half-space, in IPC
// contact through point-triangle pairs in both directions and edge-edge pairs.  Reference call sites: ball_rolling_uipc.py:71-92 (ground_height 0.001,
// d_hat 5e-4, the ball as AffineBodyConstitutionCfg), uipc_object.py:62-74,456-466 (m_kappa 100 MPa, kinematic = False),
// uipc_sim.py:192-201 (ground + one default contact model).  The arithmetic is libuipc's, which is not in the reference tree: this
// follows oracle/abd_oracle.py (PARITY UNPINNED) term by term -
//   * the ball's 12 degrees of freedom are FOUR MORE ROWS of the env's state: y = (x_0 .. x_{V-1}, p, c_1, c_2, c_3), a surface point
//     of the ball is Y_b . (p, c_1, c_2, c_3) with Y_b = (1, X_b); inertia 1/2 (q - q~)^T (S (x) I) (q - q~), orthogonality energy
//     kappa vol |A^T A - I|^2 with its Gauss-Newton Hessian 4 kappa vol [delta_mn A A^T + c_n c_m^T];
//   * barrier kappa w b(d / d_hat) on every pair (pad surface vertex, ball triangle) and (ball vertex, pad surface triangle) closer
//     than d_hat, on every EDGE-EDGE pair (pad surface edge, ball edge) closer than d_hat (segment-segment distance, weight = the mean
//     of the two edge areas, IPC's mollifier of nearly parallel pairs), and of the ground against the surface vertices of both bodies;
//     Hessians b'' grad d grad d^T;
//   * lagged Coulomb friction of every one of those contacts (the cfg's one default contact model, US:103-124 / 192-201): normal force,
//     normal and barycentric weights frozen at the state the step starts from, sliding measured relative to it (IPC's lag);
//   * matrix-free PCG, every vector in LDS (tet rows by ds_add_f64, the tet state recomputed from x); preconditioner = the pad's vertex
//     chains (block-tridiagonal LDL^T through the thickness; a chain of one vertex = a 3 x 3 block) + the additive coarse correction of
//     the pad's coarse space (tacex_fem_set_coarse_space) + the EXACT inverse of the 12 x 12 ball block;
//   * step bound: ground gaps (linear), additive CCD on the listed pairs, no surface point further than 0.9 d_hat per iteration;
//   * backtracking line search on the plain incremental potential; convergence on the unscaled direction: velocity_tol * dt on the
//     position rows AND transrate_tol * dt on the ball's affine rows (UipcSimCfg.newton, uipc_sim.py:62-66).
// One workgroup (512 threads) per env, the whole Newton loop of a time step in ONE launch.  x and the six PCG vectors live in LDS for the
// whole launch; while an iteration's system is set up the idle PCG vectors hold the gradient, the pad's 3 x 3 blocks and, in the line
// search, the candidate; the workspace block (L2 / HBM) keeps the pair list and records, the ball's surface points and the direction for
// the step bound.  Pair candidates are rebuilt once per Newton iteration at reach 2.8 d_hat, which no pair
// outside can cross into d_hat within one bounded step, so the line search's energies are exact.  (The CU-resident kernel of the
// prescribed-indenter scenes keeps ALL per-vertex state in registers and is at its register limit: this scene got a kernel of its own.)

struct BallDev {
  int nv = 0, nt = 0, npt = 0, nsv = 0;
  const double* Y = nullptr;      // (nv,4) [1, X]
  const int* tri = nullptr;       // (nt,3)
  const double* area = nullptr;   // (nv) vertex areas
  const int* ptri = nullptr;      // (npt,3) pad surface triangles
  const int* psv = nullptr;       // (nsv) pad surface vertices
  const double* parea = nullptr;  // (V) pad vertex areas (0: interior)
  double S[16] = {};              // 4 x 4 moment matrix
  double kv = 0.0, gh = 0.0, dhat = 0.0, kappa = 0.0;
  // edge-edge pairs: pad surface edges x ball edges (unique edges of the two triangle lists), the area each edge stands for (a third of its
  // triangles' rest areas) and its squared rest length (the mollifier's threshold); ee = 0 switches the pair kind off
  int npe = 0, nbe = 0, ee = 0;
  const int* pedge = nullptr;      // (npe,2)
  const double* pearea = nullptr;  // (npe)
  const double* pelen2 = nullptr;  // (npe)
  const int* bedge = nullptr;      // (nbe,2)
  const double* bearea = nullptr;  // (nbe)
  const double* belen2 = nullptr;  // (nbe)
  int ground = 0;
  int kinematic = 0;  // AffineBodyConstitutionCfg.kinematic (uipc_object.py:70-73, 463-466: `is_fixed`): the body's rows are not unknowns - the
                      // caller moves q between steps, the pad sees it through the pairs (both ways) and their friction
};

constexpr int kBallMaxPairs = 4096;   // listed candidate pairs per env and Newton iteration
constexpr int kBallMaxActive = 1024;  // pairs inside d_hat at the iteration's state
constexpr int kBallRec = 14;          // doubles per active record
constexpr int kBallMaxCand = 512;     // candidate pad vertices / pad triangles / ball vertices per env
constexpr double kBallReach = 2.0;    // additive CCD on pairs closer than kBallReach * d_hat
constexpr double kBallKeep = 0.1;     // ... which may keep this fraction of their gap
constexpr int kBallMaxFric = 1024;    // lagged friction contacts per env and time step (pairs + ground)
constexpr int kBallFlagOverflow = 16; // step_info flag: a candidate / pair list overflowed (the scene is outside what this slice handles)

// workspace of one env (doubles): ground curvature V | xb 3nv | xbc 3nv | dxb 3nv | ball triangle spheres 4nt | pair list (ints)
//   kBallMaxPairs / 2 | active records | friction records + their Hessians.  (Everything per-vertex lives in LDS.)
// dynamic LDS: x (V,3) | p (V + 4,3) | H.p accumulators = H.p (V,3) | z (V,3) | r (V + 4,3) | d (V + 4,3) (doubles) || chain factors (V,15)
// (floats) || chain successor / predecessor (V each, u16): every vector of the PCG loop, 104 KB at 495 vertices
__host__ __device__ inline size_t ball_lds_bytes(int V) {
  return ((((size_t)18 * V + 36) * sizeof(double) + (size_t)15 * V * sizeof(float) + (size_t)2 * V * sizeof(unsigned short)) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t ball_ws_doubles(int V, int T, int nv, int nt) {
  (void)T;
  return (size_t)V + (size_t)9 * nv + (size_t)4 * nt + kBallMaxPairs / 2 + (size_t)kBallMaxActive * kBallRec +
         (size_t)kBallMaxFric * (kBallRec + 6);  // lagged friction records + their Hessians at the iteration's state
}

// closest point of triangle (a, b, c) to p: barycentric coordinates, distance, unit vector from the closest point to p
// (Ericson 5.1.5, regions in the book's order - oracle/abd_oracle.py point_triangle)
__device__ __forceinline__ void pt_closest(const double p[3], const double a[3], const double b[3], const double c[3], double beta[3], double& d,
                                           double n[3]) {
  const double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
  const double ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
  const double d1 = ab[0] * ap[0] + ab[1] * ap[1] + ab[2] * ap[2], d2 = ac[0] * ap[0] + ac[1] * ap[1] + ac[2] * ap[2];
  const double bp[3] = {p[0] - b[0], p[1] - b[1], p[2] - b[2]};
  const double d3 = ab[0] * bp[0] + ab[1] * bp[1] + ab[2] * bp[2], d4 = ac[0] * bp[0] + ac[1] * bp[1] + ac[2] * bp[2];
  const double cp[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
  const double d5 = ab[0] * cp[0] + ab[1] * cp[1] + ab[2] * cp[2], d6 = ac[0] * cp[0] + ac[1] * cp[1] + ac[2] * cp[2];
  const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
  double s, u;
  if (d1 <= 0.0 && d2 <= 0.0) { s = 0.0; u = 0.0; }
  else if (d3 >= 0.0 && d4 <= d3) { s = 1.0; u = 0.0; }
  else if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { s = d1 / (d1 - d3); u = 0.0; }
  else if (d6 >= 0.0 && d5 <= d6) { s = 0.0; u = 1.0; }
  else if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { s = 0.0; u = d2 / (d2 - d6); }
  else if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) { u = (d4 - d3) / ((d4 - d3) + (d5 - d6)); s = 1.0 - u; }
  else { const double den = 1.0 / (va + vb + vc); s = vb * den; u = vc * den; }
  beta[0] = 1.0 - s - u; beta[1] = s; beta[2] = u;
  const double r0 = ap[0] - s * ab[0] - u * ac[0], r1 = ap[1] - s * ab[1] - u * ac[1], r2 = ap[2] - s * ab[2] - u * ac[2];
  d = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
  const double id = d > 0.0 ? 1.0 / d : 0.0;
  n[0] = r0 * id; n[1] = r1 * id; n[2] = r2 * id;
}

// additive CCD of one point-triangle pair (oracle/abd_oracle.py accd_point_triangle): largest t <= t_max keeping >= keep * d(0)
__device__ double accd_pt(const double p[3], const double tr[9], const double dp_in[3], const double dtr_in[9], double t_max) {
  double dp[3], dtr[9], mean[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) mean[i] = (dp_in[i] + dtr_in[i] + dtr_in[3 + i] + dtr_in[6 + i]) * 0.25;
#pragma unroll
  for (int i = 0; i < 3; ++i) { dp[i] = dp_in[i] - mean[i]; dtr[i] = dtr_in[i] - mean[i]; dtr[3 + i] = dtr_in[3 + i] - mean[i]; dtr[6 + i] = dtr_in[6 + i] - mean[i]; }
  double lt = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) lt = fmax(lt, sqrt(dtr[3 * k] * dtr[3 * k] + dtr[3 * k + 1] * dtr[3 * k + 1] + dtr[3 * k + 2] * dtr[3 * k + 2]));
  const double l = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]) + lt;
  if (!(l > 0.0)) return t_max;
  auto dist = [&](double t) {
    double q[3], a[3], b[3], c[3], be[3], d, n[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { q[i] = p[i] + t * dp[i]; a[i] = tr[i] + t * dtr[i]; b[i] = tr[3 + i] + t * dtr[3 + i]; c[i] = tr[6 + i] + t * dtr[6 + i]; }
    pt_closest(q, a, b, c, be, d, n);
    return d;
  };
  const double d0 = dist(0.0), g = kBallKeep * d0;
  double t = 0.0, tl = (1.0 - kBallKeep) * d0 / l;
  for (int it = 0; it < 64; ++it) {
    const double d = dist(t + tl);
    if (t > 0.0 && d < g) break;
    t += tl;
    if (t >= t_max) return t_max;
    tl = kCcdSlack * d / l;
  }
  return t;
}

// closest points of segments a0-a1 and b0-b1 (Ericson 5.1.9 - oracle/abd_oracle.py segment_segment): parameters, distance, unit vector from
// the point on b to the point on a; the distance's gradient is (1-s) n, s n on a's end points and -(1-t) n, -t n on b's
__device__ __forceinline__ void ee_closest(const double a0[3], const double a1[3], const double b0[3], const double b1[3], double& s, double& t, double& d,
                                           double n[3]) {
  const double d1[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]}, d2[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
  const double r[3] = {a0[0] - b0[0], a0[1] - b0[1], a0[2] - b0[2]};
  const double a = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2], e = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
  const double f = d2[0] * r[0] + d2[1] * r[1] + d2[2] * r[2], c = d1[0] * r[0] + d1[1] * r[1] + d1[2] * r[2];
  const double bq = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
  const double den = a * e - bq * bq;
  s = den > 0.0 ? fmin(fmax((bq * f - c * e) / den, 0.0), 1.0) : 0.0;
  t = (bq * s + f) / e;
  if (t < 0.0) { t = 0.0; s = fmin(fmax(-c / a, 0.0), 1.0); }
  else if (t > 1.0) { t = 1.0; s = fmin(fmax((bq - c) / a, 0.0), 1.0); }
  const double w0 = r[0] + s * d1[0] - t * d2[0], w1 = r[1] + s * d1[1] - t * d2[1], w2 = r[2] + s * d1[2] - t * d2[2];
  d = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
  const double id = d > 0.0 ? 1.0 / d : 0.0;
  n[0] = w0 * id; n[1] = w1 * id; n[2] = w2 * id;
}

// IPC's mollifier of nearly parallel edge pairs in c = |e_a x e_b|^2 (Li et al. 2020 eq. 24 - oracle edge_mollifier): m, dm/dc; u = e_a x e_b
__device__ __forceinline__ void ee_mollifier(const double a0[3], const double a1[3], const double b0[3], const double b1[3], double eps, double& mol, double& dm,
                                             double e1[3], double e2[3], double u[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) { e1[i] = a1[i] - a0[i]; e2[i] = b1[i] - b0[i]; }
  u[0] = e1[1] * e2[2] - e1[2] * e2[1]; u[1] = e1[2] * e2[0] - e1[0] * e2[2]; u[2] = e1[0] * e2[1] - e1[1] * e2[0];
  const double r = (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) / eps;
  if (r < 1.0) { mol = (2.0 - r) * r; dm = 2.0 * (1.0 - r) / eps; }
  else { mol = 1.0; dm = 0.0; }
}

// additive CCD of one edge-edge pair (oracle accd_edge_edge); ea, eb, dea, deb: the two end points of each edge, row-major (2,3)
__device__ double accd_ee(const double ea[6], const double eb[6], const double dea_in[6], const double deb_in[6], double t_max) {
  double da[6], db[6], mean[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) mean[i] = (dea_in[i] + dea_in[3 + i] + deb_in[i] + deb_in[3 + i]) * 0.25;
#pragma unroll
  for (int i = 0; i < 3; ++i) { da[i] = dea_in[i] - mean[i]; da[3 + i] = dea_in[3 + i] - mean[i]; db[i] = deb_in[i] - mean[i]; db[3 + i] = deb_in[3 + i] - mean[i]; }
  const double la = fmax(sqrt(da[0] * da[0] + da[1] * da[1] + da[2] * da[2]), sqrt(da[3] * da[3] + da[4] * da[4] + da[5] * da[5]));
  const double lb = fmax(sqrt(db[0] * db[0] + db[1] * db[1] + db[2] * db[2]), sqrt(db[3] * db[3] + db[4] * db[4] + db[5] * db[5]));
  const double l = la + lb;
  if (!(l > 0.0)) return t_max;
  auto dist = [&](double t) {
    double a0[3], a1[3], b0[3], b1[3], s, u, d, n[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { a0[i] = ea[i] + t * da[i]; a1[i] = ea[3 + i] + t * da[3 + i]; b0[i] = eb[i] + t * db[i]; b1[i] = eb[3 + i] + t * db[3 + i]; }
    ee_closest(a0, a1, b0, b1, s, u, d, n);
    return d;
  };
  const double d0 = dist(0.0), g = kBallKeep * d0;
  double t = 0.0, tl = (1.0 - kBallKeep) * d0 / l;
  for (int it = 0; it < 64; ++it) {
    const double d = dist(t + tl);
    if (t > 0.0 && d < g) break;
    t += tl;
    if (t >= t_max) return t_max;
    tl = kCcdSlack * d / l;
  }
  return t;
}

__device__ __forceinline__ void barrier3(double s, double& b, double& b1, double& b2) {  // b(s), b'(s), b''(s) of the dimensionless barrier
  if (!(s > 0.0)) { b = INFINITY; b1 = 0.0; b2 = 0.0; return; }
  if (s >= 1.0) { b = 0.0; b1 = 0.0; b2 = 0.0; return; }
  const double ln = log(s), q = s - 1.0;
  b = -q * q * ln;
  b1 = -2.0 * q * ln - q * q / s;
  b2 = -2.0 * ln - 4.0 * q / s + q * q / (s * s);
}

// mode 0: Newton loop of a time step; mode 1: energy and gradient at (x, q) only (terms entry point of the C ABI: tests)
#ifndef TACEX_BALL_WG_PER_CU
#define TACEX_BALL_WG_PER_CU 2  // waves per SIMD the kernel is compiled for (2 = one env per CU; A/B: 4 = two envs co-resident at <= 128 VGPRs, profiles/r06_experiments.md section 5)
#endif
template <int NT_>
__global__ __launch_bounds__(NT_, NT_ >= 512 ? TACEX_BALL_WG_PER_CU : 1) void fem_ball_newton_kernel(FemDev m, BallDev bd, double* xg, const double* xtg, double* qg, const double* qtg,
                                                              const uint8_t* consg, const double* aimg, double* wsg, int pcg_max_iter,
                                                              double pcg_tol_rate, int ls_max_iter, int max_newton, double dx_tol, double dc_tol,
                                                              double* step_info, int mode, double* e_out, double* g_out, const double* xprevg,
                                                              const double* qprevg, const int* env_order, const double* blkg) {
  extern __shared__ __attribute__((aligned(16))) double ball_lds[];  // x (V,3) | p (V + 4,3) | H.p accumulators (V,3): ball_lds_bytes()
  __shared__ double sh[17], sh2[16];
  __shared__ double gb[12], Bm[144], B0[144], Lc[144], Bi[144], YY[16], qs[12], qts[12], rhs12[12];
  __shared__ double crc[3 * kFemMaxCoarse], cyc[3 * kFemMaxCoarse];  // coarse residual / correction of the two-level preconditioner
  __shared__ double Hw[NT_ / 64 * 12];  // per-wave partial sums of the ball rows of H.p over the pair / friction records (one copy: 61 records x 12 atomic adds on 12 addresses)
  __shared__ double cpart[6 * kFemMaxCoarse];                        // ... the two halves of the coarse solve's sums
  __shared__ double qps[12], Hpq[12], zq[12];  // ... | the ball rows of H.p and of z  // the ball rows the time step started from (friction slides relative to them)
  __shared__ int n_cpv, n_cpt, n_cpe, n_cbv, n_cbt, n_cbe, n_pairs, n_act, n_fric, s_flags;
  __shared__ unsigned short cpv[kBallMaxCand], cpt[kBallMaxCand], cpe[kBallMaxCand];  // candidate pad vertices / triangles / edges (indices < 32768)
  __shared__ unsigned short cbv[kBallMaxCand], cbt[kBallMaxCand], cbe[kBallMaxCand];  // candidate ball vertices / triangles / edges
  // env_order: envs sorted by the solver work of their previous step, heaviest first (fem_env_order_kernel): a shard brings two envs
  // per CU, the launch ends with whatever the last-started ones need
  const int b = env_order ? env_order[blockIdx.x] : (int)blockIdx.x;
  constexpr int NT = NT_, NWV = NT_ / 64;  // threads / waves per env: 512 / 8 (two waves per SIMD, 256 VGPRs each) or 256 / 4 (one wave per SIMD with the whole register file)
  const int tid = threadIdx.x;
  const int V = m.V, T = m.T, nv = bd.nv, nt = bd.nt, VN = V + 4;
  const size_t o = (size_t)b * V * 3;
  double* x = xg + o;
  const double* xt = xtg + o;
  double* q = qg + (size_t)b * 12;
  const double* qt = qtg + (size_t)b * 12;
  const uint8_t* cons = consg ? consg + (size_t)b * V : nullptr;
  const double* aim = aimg ? aimg + o : nullptr;
  double* ws = wsg + (size_t)b * ball_ws_doubles(V, T, nv, nt);
  double* cbp = ws;                    // (V) ground curvature of the pad vertices at x (dt^2-scaled)
  double* xb = cbp + V;                // (nv,3) ball surface points at x
  double* xbc = xb + (size_t)3 * nv;   // ... at the candidate
  double* dxb = xbc + (size_t)3 * nv;  // ... their displacement along the Newton direction
  double* bts = dxb + (size_t)3 * nv;  // (nt,4) bounding sphere of every ball triangle at x
  int* plist = reinterpret_cast<int*>(bts + (size_t)4 * nt);  // (kBallMaxPairs) kind << 30 | point << 15 | triangle
  double* arec = reinterpret_cast<double*>(plist + kBallMaxPairs);
  double* frec = arec + (size_t)kBallMaxActive * kBallRec;  // lagged friction contacts of the step: lam | n (3) | ball coefficients (4) | pad coefficients (3) | pad rows (3 ints)
  double* fM = frec + (size_t)kBallMaxFric * kBallRec;      // their 3 x 3 Hessians at x (6 each, dt^2 mu lam [a T + (b - a) t t^T])
  // LAGGED COULOMB FRICTION of every contact (Li et al. 2020 eq. 18-20; US:103-124 friction ratio / eps_velocity, tacex_fem_set_friction): the
  // contacts of the state the step STARTS from - active pairs of both kinds, ground contacts of both bodies - are frozen as (normal force,
  // normal, coefficients of the relative displacement) in the first Newton iteration (x = x_n there) and slide relative to that state
  const double* xprev = xprevg ? xprevg + o : nullptr;
  const bool fric = m.fric_mu > 0.0 && xprev != nullptr && qprevg != nullptr;
  const double f_eps = m.fric_eps, f_mu = m.fric_mu;
  double* xs = ball_lds;           // (V,3) x of the iteration: the tet state is recomputed from it in every H.p (no cached F in HBM)
  double* ps = xs + 3 * V;         // (V + 4,3) PCG direction
  double* acc = ps + 3 * VN;       // (V,3) per-vertex sums of the tets' rows (ds_add_f64)
  double* rsl = acc + 3 * V;       // (V,3) z: the chain solve's r, then y, then z (in place), + the coarse correction
  double* rL = rsl + 3 * V;        // (V + 4,3) PCG residual
  double* dL = rL + 3 * VN;        // (V + 4,3) PCG solution (the Newton direction)
  // While an iteration's system is SET UP the PCG vectors are idle: the gradient (3 (V + 4)) and the pad's diagonal blocks (V,9) are built in
  // their place - hundreds of f64 atomic adds per iteration (pair / friction rows) then hit LDS instead of memory, and the chain
  // factorisation reads its blocks from LDS.  vg = dL's slot (rL = -vg, dL = 0 is the hand-over, element by element); Dinv = acc | z | the pad
  // rows of r, all read for the last time by the chain factorisation, one barrier before r is written.
  double* const vg = dL;
  double* const ycl = ps;  // line-search candidate (pad rows): the PCG direction's slot, idle between the PCG loop and the next one
  double* const Dinv = acc;
  float* cf = reinterpret_cast<float*>(dL + 3 * VN);                      // (V,15) chain factors: S^-1 (upper triangle, 6) | G (9)
  unsigned short* cnx = reinterpret_cast<unsigned short*>(cf + 15 * V);   // (V) chain successor, 0xffff = none
  unsigned short* cpr = cnx + V;                                           // (V) predecessor
  // elastic preconditioner blocks of the step: D (upper triangle) | E = A(v, next(v)) per vertex, assembled for all envs by
  // fem_assemble_blocks_kernel ahead of this launch at the state the step starts from ((V,16) per env; nullptr: mass blocks only)
  const double* blk = blkg ? blkg + (size_t)b * 16 * V : nullptr;
  const int nch = m.ch_next ? m.nch : V;
  for (int v = tid; v < V; v += NT) {
    cnx[v] = (unsigned short)(m.ch_next ? m.ch_next[v] : -1);
    cpr[v] = (unsigned short)(m.ch_prev ? m.ch_prev[v] : -1);
  }
  const double dt2 = m.dt * m.dt, dhat = bd.dhat, kk = dt2 * bd.kappa;
  const double L = dhat * (1.0 + kCcdSlack * kBallReach), R = kBallReach * dhat;

  if (tid < 12) { qs[tid] = q[tid]; qts[tid] = qt[tid]; qps[tid] = fric ? qprevg[(size_t)b * 12 + tid] : 0.0; }
  if (tid == 0) { s_flags = 0; n_fric = 0; }
  __syncthreads();

  auto ball_points = [&](const double* qq, double* out) {  // out (nv,3) = Y qq; qq in LDS
    for (int k = tid; k < nv; k += NT) {
      const double y0 = bd.Y[k * 4], y1 = bd.Y[k * 4 + 1], y2 = bd.Y[k * 4 + 2], y3 = bd.Y[k * 4 + 3];
#pragma unroll
      for (int i = 0; i < 3; ++i) out[k * 3 + i] = y0 * qq[i] + y1 * qq[3 + i] + y2 * qq[6 + i] + y3 * qq[9 + i];
    }
  };
  auto fric_u = [&](const double* rc, const double* xx, const double* qq, double u[3]) {  // tangential relative displacement since the step's start
    const int* ri = reinterpret_cast<const int*>(rc + 11);
    double rel[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
      for (int i = 0; i < 3; ++i) rel[i] += rc[4 + a4] * (qq[a4 * 3 + i] - qps[a4 * 3 + i]);
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (ri[r] >= 0)
#pragma unroll
        for (int i = 0; i < 3; ++i) rel[i] += rc[8 + r] * (xx[ri[r] * 3 + i] - xprev[ri[r] * 3 + i]);
    const double rn = rel[0] * rc[1] + rel[1] * rc[2] + rel[2] * rc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) u[i] = rel[i] - rn * rc[1 + i];
  };
  // energy of the state (xx pad vertices in global memory, qq ball rows in LDS, xbb its surface points): every term of
  // oracle/abd_oracle.py BallScene.energy; the pairs are those of the iteration's list
  auto energy = [&](const double* xx, const double* qq, const double* xbb) -> double {
    double e = 0.0;
    for (int t = tid; t < T; t += NT) {
      int v[4];
      double Di[9], F[9];
      load_tet(m, t, v, Di);
      deformation_gradient(xx, v, Di, F);
      TetState s;
      tet_state(m, F, s);
      e += dt2 * m.vol[t] * psi_of(m, s);
    }
    for (int v = tid; v < V; v += NT) {
      const double mv = m.mass[v];
      double qd = 0.0, qc = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double d = xx[v * 3 + i] - xt[v * 3 + i];
        qd += d * d;
        if (cons && cons[v]) { const double c = xx[v * 3 + i] - aim[v * 3 + i]; qc += c * c; }
      }
      e += 0.5 * mv * qd + 0.5 * m.strength * mv * qc;
      const double w = bd.parea[v];
      if (bd.ground && w > 0.0) {
        double bb, b1, b2;
        barrier3((xx[v * 3 + 2] - bd.gh) / dhat, bb, b1, b2);
        e += kk * w * bb;
      }
    }
    if (bd.ground)
      for (int k = tid; k < nv; k += NT) {
        double bb, b1, b2;
        barrier3((xbb[k * 3 + 2] - bd.gh) / dhat, bb, b1, b2);
        e += kk * bd.area[k] * bb;
      }
    if (tid == 0) {
      double ei = 0.0;
      for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 4; ++c) {
          double dd = 0.0;
#pragma unroll
          for (int i = 0; i < 3; ++i) dd += (qq[a * 3 + i] - qts[a * 3 + i]) * (qq[c * 3 + i] - qts[c * 3 + i]);
          ei += bd.S[a * 4 + c] * dd;
        }
      double eo = 0.0;
      for (int k = 0; k < 3; ++k)
        for (int l = 0; l < 3; ++l) {
          const double r = qq[3 + k * 3] * qq[3 + l * 3] + qq[4 + k * 3] * qq[4 + l * 3] + qq[5 + k * 3] * qq[5 + l * 3] - (k == l ? 1.0 : 0.0);
          eo += r * r;
        }
      e += 0.5 * ei + dt2 * bd.kv * eo;
    }
    const int np = n_pairs;
    for (int k = tid; k < np; k += NT) {
      const unsigned code = (unsigned)plist[k];
      const int kind = (int)(code >> 30), pi = (int)((code >> 15) & 0x7fff), tj = (int)(code & 0x7fff);
      double d, w;
      if (kind == 2) {
        const int* ea = bd.pedge + pi * 2;
        const int* eb = bd.bedge + tj * 2;
        double a0[3], a1[3], b0[3], b1[3], s, t, n[3], mol, dm, e1[3], e2[3], u[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { a0[i] = xx[ea[0] * 3 + i]; a1[i] = xx[ea[1] * 3 + i]; b0[i] = xbb[eb[0] * 3 + i]; b1[i] = xbb[eb[1] * 3 + i]; }
        ee_closest(a0, a1, b0, b1, s, t, d, n);
        ee_mollifier(a0, a1, b0, b1, 1e-3 * bd.pelen2[pi] * bd.belen2[tj], mol, dm, e1, e2, u);
        w = 0.5 * (bd.pearea[pi] + bd.bearea[tj]) * mol;
      } else {
        double p3[3], a[3], bq[3], c[3], be[3], n[3];
        if (kind == 0) {
          const int* tr = bd.tri + tj * 3;
#pragma unroll
          for (int i = 0; i < 3; ++i) { p3[i] = xx[pi * 3 + i]; a[i] = xbb[tr[0] * 3 + i]; bq[i] = xbb[tr[1] * 3 + i]; c[i] = xbb[tr[2] * 3 + i]; }
          w = bd.parea[pi];
        } else {
          const int* tr = bd.ptri + tj * 3;
#pragma unroll
          for (int i = 0; i < 3; ++i) { p3[i] = xbb[pi * 3 + i]; a[i] = xx[tr[0] * 3 + i]; bq[i] = xx[tr[1] * 3 + i]; c[i] = xx[tr[2] * 3 + i]; }
          w = bd.area[pi];
        }
        pt_closest(p3, a, bq, c, be, d, n);
      }
      if (d < dhat) {
        double bb, b1, b2;
        barrier3(d / dhat, bb, b1, b2);
        if (w > 0.0) e += kk * w * bb;  // (w = 0: an exactly parallel edge pair - mollified away, whatever its distance)
      }
    }
    if (fric) {
      const int nf = n_fric;
      for (int k = tid; k < nf; k += NT) {
        const double* rc = frec + (size_t)k * kBallRec;
        double u[3];
        fric_u(rc, xx, qq, u);
        const double yv = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        e += dt2 * f_mu * rc[0] * (yv < f_eps ? -yv * yv * yv / (3.0 * f_eps * f_eps) + yv * yv / f_eps + f_eps / 3.0 : yv);
      }
    }
    return block_sum(e, sh);
  };
  // block-wide sum with ONE barrier (two alternating rows of wave partials: the row written two calls ago cannot still be read, a
  // barrier lies in between) - the PCG loop runs two of these per iteration
  int bphase = 0;
  auto bsum = [&](double v) -> double {
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2, 64);
    double* row = sh2 + 8 * (bphase & 1);
    ++bphase;
    if ((tid & 63) == 0) row[tid >> 6] = v;
    __syncthreads();
    double sv = 0.0;
#pragma unroll
    for (int w8 = 0; w8 < NWV; ++w8) sv += row[w8];
    return sv;
  };
  const bool coarse = m.nc > 0 && m.cn_off && m.ac_inv && (mode & 2) == 0;  // (mode bit 1: block Jacobi alone, A/B)
  // z = M^-1 r, everything in LDS (r = rL; z = rsl on the pad rows, zq on the ball rows): chain solve z = L^-T S^-1 L^-1 r in place (down the
  // chain y_i = r_i - G_{i-1}^T y_{i-1}, back up z_i = S_i^-1 y_i - G_i z_{i+1}) + the additive coarse correction P A_c^-1 P^T r of the
  // pad's coarse space (restriction by ds_add_f64 from the vertices' own 8 (node, weight) pairs) + the exact inverse of the ball block.
  // Returns r . z.
#ifdef TACEX_BALL_CLOCK  // debug build: cycles inside the PCG loop (sweep | vertex + records | dot + update | restriction + chains | coarse solve | prolongation + sum)
  long long pck[7] = {0, 0, 0, 0, 0, 0, 0}, pt0 = 0;
#define PCG_TICK0() do { pt0 = __builtin_readcyclecounter(); } while (0)
#define PCG_TICK(k) do { const long long n_ = __builtin_readcyclecounter(); pck[k] += n_ - pt0; pt0 = n_; } while (0)
#else
#define PCG_TICK0() do { } while (0)
#define PCG_TICK(k) do { } while (0)
#endif
  auto chain_solve = [&](int ch) {  // block-tridiagonal L D L^T solve along one vertex chain, in place in rsl (factors cf in LDS)
    int v = m.ch_next ? m.ch_head[ch] : ch, last = v;
    double y[3] = {rsl[v * 3], rsl[v * 3 + 1], rsl[v * 3 + 2]};
    while (true) {
      last = v;
      const int n = cnx[v] == 0xffff ? -1 : (int)cnx[v];
      if (n < 0) break;
      const float* g = cf + v * 15 + 6;
      const double y0 = y[0], y1 = y[1], y2 = y[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) y[k] = rsl[n * 3 + k] - ((double)g[k] * y0 + (double)g[3 + k] * y1 + (double)g[6 + k] * y2);
      rsl[n * 3] = y[0]; rsl[n * 3 + 1] = y[1]; rsl[n * 3 + 2] = y[2];
      v = n;
    }
    v = last;
    double zn[3] = {0, 0, 0};
    while (true) {
      const float* f = cf + v * 15;
      const double y0 = rsl[v * 3], y1 = rsl[v * 3 + 1], y2 = rsl[v * 3 + 2];
      double zz[3];
      zz[0] = (double)f[0] * y0 + (double)f[1] * y1 + (double)f[2] * y2;
      zz[1] = (double)f[1] * y0 + (double)f[3] * y1 + (double)f[4] * y2;
      zz[2] = (double)f[2] * y0 + (double)f[4] * y1 + (double)f[5] * y2;
#pragma unroll
      for (int i = 0; i < 3; ++i) zz[i] -= (double)f[6 + i * 3] * zn[0] + (double)f[7 + i * 3] * zn[1] + (double)f[8 + i * 3] * zn[2];
      rsl[v * 3] = zz[0]; rsl[v * 3 + 1] = zz[1]; rsl[v * 3 + 2] = zz[2];
      zn[0] = zz[0]; zn[1] = zz[1]; zn[2] = zz[2];
      const int pv = cpr[v] == 0xffff ? -1 : (int)cpr[v];
      if (pv < 0) break;
      v = pv;
    }
  };
  // z = M^-1 r (r in rL, z in rsl | zq); returns r . z.  Phases: (A) copy + restriction r_c = P^T r; (B) coarse solve on waves 0-5 WHILE waves 6-7
  // solve the chains; (C) the coarse solve's two halves; (D) prolongation + r . z.
  auto precondition = [&]() -> double {
    const int nc3 = 3 * m.nc;
    const int lane = tid & 63, wave = tid >> 6;
    PCG_TICK0();
    for (int k = tid; k < 3 * V; k += NT) rsl[k] = rL[k];
    if (coarse) {  // restriction, node by node over the node's vertex list (G lanes per node, as coarse_correct(); the first cut added 12 000 LDS
                   // atomics onto 180 addresses per application)
      int G = 1;
      while (2 * G <= NT / m.nc && 2 * G <= 64) G *= 2;
      const int node = tid / G, jn = tid - node * G;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      if (node < m.nc) {
        const int e1 = m.cn_off[node + 1];
        for (int e = m.cn_off[node] + jn; e < e1; e += G) {
          const int v0 = m.cn_vtx[e];
          const double w0 = m.cn_w[e];
          a0 += w0 * rL[v0 * 3]; a1 += w0 * rL[v0 * 3 + 1]; a2 += w0 * rL[v0 * 3 + 2];
        }
      }
      for (int o2 = G >> 1; o2 > 0; o2 >>= 1) { a0 += __shfl_xor(a0, o2, 64); a1 += __shfl_xor(a1, o2, 64); a2 += __shfl_xor(a2, o2, 64); }
      if (node < m.nc && jn == 0) { crc[node * 3] = a0; crc[node * 3 + 1] = a1; crc[node * 3 + 2] = a2; }
    }
    __syncthreads();
    PCG_TICK(3);
    double part = 0.0;
    if (tid < 12) {
      double zz = 0.0;
#pragma unroll
      for (int k = 0; k < 12; ++k) zz += Bi[tid * 12 + k] * rL[V * 3 + k];
      zq[tid] = zz;
      part += rL[V * 3 + tid] * zz;
    }
    constexpr int MW = NWV >= 8 ? 6 : 3, HALVES = MW / 3;  // waves of the coarse solve (three output chunks x one or two halves of the sum); the rest solve the chains
    if (coarse && NWV > MW) {
      // y_c = A_c^-1 r_c, a (3 nc)^2 <= 192^2 f64 matrix in L2.  The matrix is symmetric: lane = OUTPUT index, the loop runs down a COLUMN
      // block - every load instruction reads 512 contiguous bytes, eight of them in flight, no cross-lane reduction.  Six waves: three
      // output chunks of 64 x two halves of the sum.  (A thread walking its own row: 32 of a PCG iteration's 110 kcycles with the 60-node
      // grid - 180 dependent steps, 64 cache lines per load instruction; a wave per row with shuffles: 36.)
      if (wave < MW) {
        const int r = (wave % 3) * 64 + lane, jp = wave / 3;
        const int j0 = jp * (192 / HALVES), j1 = min(nc3, j0 + 192 / HALVES);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (r < nc3) {
          int jj = j0;
          for (; jj + 8 <= j1; jj += 8) {
            double a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = m.ac_inv[(size_t)(jj + u) * nc3 + r];
            s0 += a[0] * crc[jj] + a[4] * crc[jj + 4];
            s1 += a[1] * crc[jj + 1] + a[5] * crc[jj + 5];
            s2 += a[2] * crc[jj + 2] + a[6] * crc[jj + 6];
            s3 += a[3] * crc[jj + 3] + a[7] * crc[jj + 7];
          }
          for (; jj < j1; ++jj) s0 += m.ac_inv[(size_t)jj * nc3 + r] * crc[jj];
          cpart[jp * 3 * kFemMaxCoarse + r] = (s0 + s1) + (s2 + s3);
        }
      } else {  // ... while the last two waves solve the chains
        for (int ch = tid - MW * 64; ch < nch; ch += NT - MW * 64) chain_solve(ch);
      }
      __syncthreads();
      for (int k = tid; k < nc3; k += NT) cyc[k] = cpart[k] + (HALVES > 1 ? cpart[3 * kFemMaxCoarse + k] : 0.0);
      __syncthreads();
      for (int k = tid; k < nc3; k += NT) part += crc[k] * cyc[k];
    } else {
      for (int ch = tid; ch < nch; ch += NT) chain_solve(ch);
      __syncthreads();
    }
    PCG_TICK(4);
    for (int v = tid; v < V; v += NT) {
      double z0 = rsl[v * 3], z1 = rsl[v * 3 + 1], z2 = rsl[v * 3 + 2];
      part += rL[v * 3] * z0 + rL[v * 3 + 1] * z1 + rL[v * 3 + 2] * z2;  // (the chain part; the coarse part of r . z is crc . cyc above)
      if (coarse) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int nd = m.cv_node[v * 8 + k];
          const double w = m.cv_w[v * 8 + k];
          z0 += w * cyc[nd * 3]; z1 += w * cyc[nd * 3 + 1]; z2 += w * cyc[nd * 3 + 2];
        }
        rsl[v * 3] = z0; rsl[v * 3 + 1] = z1; rsl[v * 3 + 2] = z2;
      }
    }
    const double rzv = bsum(part);
    PCG_TICK(5);
    return rzv;
  };

#ifdef TACEX_BALL_CLOCK  // debug build: cycles of the phases of a step, printed by env 0 (scripts/r06/ball_clock.sh)
  long long bck[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bt0 = __builtin_readcyclecounter();
  long long sck[4] = {0, 0, 0, 0}, st0 = 0;
#define SUB_TICK0() do { st0 = __builtin_readcyclecounter(); } while (0)
#define SUB_TICK(k) do { const long long n_ = __builtin_readcyclecounter(); sck[k] += n_ - st0; st0 = n_; } while (0)
#define BALL_TICK(k) do { const long long n_ = __builtin_readcyclecounter(); bck[k] += n_ - bt0; bt0 = n_; } while (0)
#else
#define BALL_TICK(k) do { } while (0)
#define SUB_TICK0() do { } while (0)
#define SUB_TICK(k) do { } while (0)
#endif
  ball_points(qs, xb);
  for (int k = tid; k < 3 * V; k += NT) xs[k] = x[k];
  __syncthreads();
  int n_newton = 0, pcg_total = 0;
  double e_carry = 0.0;
  bool have_e = false;
  double dmax_x = INFINITY, dmax_c = INFINITY;
  for (int nit = 0; nit < max_newton; ++nit) {
    const bool take_lag = fric && (nit == 0);  // (the first iteration stands at the state the step starts from)
    // ---- element pass (as fem_newton_kernel) + bounding spheres of the ball triangles + candidate lists ----
    for (int k = tid; k < 3 * VN; k += NT) vg[k] = 0.0;
    __syncthreads();
    for (int t = tid; t < T; t += NT) {  // (the tets' rows go straight into the LDS gradient: no (12,T) array through memory, no gather)
      int v[4];
      double Di[9], F[9], r[12], g[12];
      load_tet(m, t, v, Di);
      deformation_gradient(xs, v, Di, F);
      TetState s;
      tet_state(m, F, s);
      shape_rows(Di, r);
      element_gradient(s, r, dt2 * m.vol[t], g);
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4)
#pragma unroll
        for (int i = 0; i < 3; ++i) atomicAdd(&vg[v[w4] * 3 + i], g[w4 * 3 + i]);
    }
    double rb = 0.0;  // bounding radius of the ball about p
    for (int k = tid; k < nv; k += NT) {
      const double r0 = xb[k * 3] - qs[0], r1 = xb[k * 3 + 1] - qs[1], r2 = xb[k * 3 + 2] - qs[2];
      rb = fmax(rb, sqrt(r0 * r0 + r1 * r1 + r2 * r2));
    }
    for (int t = tid; t < nt; t += NT) {
      const int* tr = bd.tri + t * 3;
      double c3[3], rt = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) c3[i] = (xb[tr[0] * 3 + i] + xb[tr[1] * 3 + i] + xb[tr[2] * 3 + i]) * (1.0 / 3.0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double r0 = xb[tr[k] * 3] - c3[0], r1 = xb[tr[k] * 3 + 1] - c3[1], r2 = xb[tr[k] * 3 + 2] - c3[2];
        rt = fmax(rt, sqrt(r0 * r0 + r1 * r1 + r2 * r2));
      }
      bts[t * 4] = c3[0]; bts[t * 4 + 1] = c3[1]; bts[t * 4 + 2] = c3[2]; bts[t * 4 + 3] = rt;
    }
    BALL_TICK(0);  // element pass, ball triangle spheres
    if (tid == 0) { n_cpv = 0; n_cpt = 0; n_cpe = 0; n_cbv = 0; n_cbt = 0; n_cbe = 0; n_pairs = 0; n_act = 0; }
    if (tid < 12) gb[tid] = 0.0;
    if (tid < 16) YY[tid] = 0.0;
    if (tid < 144) Bm[tid] = 0.0;
    rb = block_sum_max(rb, sh);
    // pad surface vertices / triangles within reach of the ball's bounding sphere; ball vertices within reach of ANY candidate pad triangle
    // are found pair by pair below
    for (int k = tid; k < bd.nsv; k += NT) {
      const int v = bd.psv[k];
      const double r0 = xs[v * 3] - qs[0], r1 = xs[v * 3 + 1] - qs[1], r2 = xs[v * 3 + 2] - qs[2];
      const double lim = rb + L;
      if (r0 * r0 + r1 * r1 + r2 * r2 < lim * lim) {
        const int s = atomicAdd(&n_cpv, 1);
        if (s < kBallMaxCand) cpv[s] = (unsigned short)v;
      }
    }
    for (int k = tid; k < bd.npt; k += NT) {
      const int* tr = bd.ptri + k * 3;
      double c3[3], rt = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) c3[i] = (xs[tr[0] * 3 + i] + xs[tr[1] * 3 + i] + xs[tr[2] * 3 + i]) * (1.0 / 3.0);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double r0 = xs[tr[j] * 3] - c3[0], r1 = xs[tr[j] * 3 + 1] - c3[1], r2 = xs[tr[j] * 3 + 2] - c3[2];
        rt = fmax(rt, sqrt(r0 * r0 + r1 * r1 + r2 * r2));
      }
      const double r0 = c3[0] - qs[0], r1 = c3[1] - qs[1], r2 = c3[2] - qs[2];
      const double lim = rb + L + rt;
      if (r0 * r0 + r1 * r1 + r2 * r2 < lim * lim) {
        const int s = atomicAdd(&n_cpt, 1);
        if (s < kBallMaxCand) cpt[s] = (unsigned short)k;
      }
    }
    if (bd.ee)
      for (int k = tid; k < bd.npe; k += NT) {
        const int* ed = bd.pedge + k * 2;
        double r2 = 0.0, h2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double mi = 0.5 * (xs[ed[0] * 3 + i] + xs[ed[1] * 3 + i]) - qs[i], hi = 0.5 * (xs[ed[1] * 3 + i] - xs[ed[0] * 3 + i]);
          r2 += mi * mi; h2 += hi * hi;
        }
        const double lim = rb + L + sqrt(h2);
        if (r2 < lim * lim) {
          const int s = atomicAdd(&n_cpe, 1);
          if (s < kBallMaxCand) cpe[s] = (unsigned short)k;
        }
      }
    __syncthreads();
    if (n_cpv > kBallMaxCand || n_cpt > kBallMaxCand || n_cpe > kBallMaxCand) { if (tid == 0) s_flags |= kBallFlagOverflow; }
    const int ncpv = min(n_cpv, kBallMaxCand), ncpt = min(n_cpt, kBallMaxCand), ncpe = min(n_cpe, kBallMaxCand);
    // the ball's side of the lists: what lies within L of the bounding box of the pad's candidates (a ball of 320 triangles / 480 edges / 162
    // vertices touches the pad with a few dozen of each: the pair tests below shrink accordingly)
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto grow = [&](int v) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { lo[i] = fmin(lo[i], xs[v * 3 + i]); hi[i] = fmax(hi[i], xs[v * 3 + i]); }
    };
    for (int k = tid; k < ncpv; k += NT) grow(cpv[k]);
    for (int k = tid; k < ncpt; k += NT) { const int* tr = bd.ptri + cpt[k] * 3; grow(tr[0]); grow(tr[1]); grow(tr[2]); }
    for (int k = tid; k < ncpe; k += NT) { const int* ed = bd.pedge + cpe[k] * 2; grow(ed[0]); grow(ed[1]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) { lo[i] = -block_sum_max(-lo[i], sh) - L; hi[i] = block_sum_max(hi[i], sh) + L; }
    for (int k = tid; k < nv; k += NT) {
      const double p0 = xb[k * 3], p1 = xb[k * 3 + 1], p2 = xb[k * 3 + 2];
      if (p0 >= lo[0] && p0 <= hi[0] && p1 >= lo[1] && p1 <= hi[1] && p2 >= lo[2] && p2 <= hi[2]) {
        const int sl = atomicAdd(&n_cbv, 1);
        if (sl < kBallMaxCand) cbv[sl] = (unsigned short)k;
      }
    }
    for (int t = tid; t < nt; t += NT) {
      const double rt = bts[t * 4 + 3];
      bool in = true;
#pragma unroll
      for (int i = 0; i < 3; ++i) in = in && bts[t * 4 + i] + rt >= lo[i] && bts[t * 4 + i] - rt <= hi[i];
      if (in) {
        const int sl = atomicAdd(&n_cbt, 1);
        if (sl < kBallMaxCand) cbt[sl] = (unsigned short)t;
      }
    }
    if (bd.ee)
      for (int k = tid; k < bd.nbe; k += NT) {
        const int* eb = bd.bedge + k * 2;
        bool in = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double a = xb[eb[0] * 3 + i], b2 = xb[eb[1] * 3 + i];
          in = in && fmax(a, b2) >= lo[i] && fmin(a, b2) <= hi[i];
        }
        if (in) {
          const int sl = atomicAdd(&n_cbe, 1);
          if (sl < kBallMaxCand) cbe[sl] = (unsigned short)k;
        }
      }
    __syncthreads();
    if (n_cbv > kBallMaxCand || n_cbt > kBallMaxCand || n_cbe > kBallMaxCand) { if (tid == 0) s_flags |= kBallFlagOverflow; }
    const int ncbv = min(n_cbv, kBallMaxCand), ncbt = min(n_cbt, kBallMaxCand), ncbe = min(n_cbe, kBallMaxCand);
    // pairs, kind 0: candidate pad vertex x candidate ball triangle
    for (int k = tid; k < ncpv * ncbt; k += NT) {
      const int v = cpv[k / ncbt], t = cbt[k - (k / ncbt) * ncbt];
      const double r0 = xs[v * 3] - bts[t * 4], r1 = xs[v * 3 + 1] - bts[t * 4 + 1], r2 = xs[v * 3 + 2] - bts[t * 4 + 2];
      const double lim = L + bts[t * 4 + 3];
      if (r0 * r0 + r1 * r1 + r2 * r2 >= lim * lim) continue;
      const int* tr = bd.tri + t * 3;
      double p3[3], a[3], bq[3], c[3], be[3], d, n[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { p3[i] = xs[v * 3 + i]; a[i] = xb[tr[0] * 3 + i]; bq[i] = xb[tr[1] * 3 + i]; c[i] = xb[tr[2] * 3 + i]; }
      pt_closest(p3, a, bq, c, be, d, n);
      if (d < L) {
        const int s = atomicAdd(&n_pairs, 1);
        if (s < kBallMaxPairs) plist[s] = (0 << 30) | (v << 15) | t;
      }
    }
    // pairs, kind 1: candidate ball vertex x candidate pad triangle
    for (int k = tid; k < ncbv * ncpt; k += NT) {
      const int bv = cbv[k / ncpt], t = cpt[k - (k / ncpt) * ncpt];
      const int* tr = bd.ptri + t * 3;
      double p3[3], a[3], bq[3], c[3], be[3], d, n[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { p3[i] = xb[bv * 3 + i]; a[i] = xs[tr[0] * 3 + i]; bq[i] = xs[tr[1] * 3 + i]; c[i] = xs[tr[2] * 3 + i]; }
      // sphere test about the triangle's first corner (its edges are bounded by the distance test itself: cheap rejection)
      const double r0 = p3[0] - a[0], r1 = p3[1] - a[1], r2 = p3[2] - a[2];
      const double e0 = bq[0] - a[0], e1 = bq[1] - a[1], e2 = bq[2] - a[2], f0 = c[0] - a[0], f1 = c[1] - a[1], f2 = c[2] - a[2];
      const double ext = sqrt(fmax(e0 * e0 + e1 * e1 + e2 * e2, f0 * f0 + f1 * f1 + f2 * f2));
      const double lim = L + ext;
      if (r0 * r0 + r1 * r1 + r2 * r2 >= lim * lim) continue;
      pt_closest(p3, a, bq, c, be, d, n);
      if (d < L) {
        const int s = atomicAdd(&n_pairs, 1);
        if (s < kBallMaxPairs) plist[s] = (1 << 30) | (bv << 15) | t;
      }
    }
    // pairs, kind 2: candidate pad edge x candidate ball edge
    for (int k = tid; k < ncpe * ncbe; k += NT) {
      const int pe = cpe[k / ncbe], be_ = cbe[k - (k / ncbe) * ncbe];
      const int* ea = bd.pedge + pe * 2;
      const int* eb = bd.bedge + be_ * 2;
      double a0[3], a1[3], b0[3], b1[3];
      double r2 = 0.0, ha = 0.0, hb = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        a0[i] = xs[ea[0] * 3 + i]; a1[i] = xs[ea[1] * 3 + i]; b0[i] = xb[eb[0] * 3 + i]; b1[i] = xb[eb[1] * 3 + i];
        const double mi = 0.5 * ((a0[i] + a1[i]) - (b0[i] + b1[i])), hai = 0.5 * (a1[i] - a0[i]), hbi = 0.5 * (b1[i] - b0[i]);
        r2 += mi * mi; ha += hai * hai; hb += hbi * hbi;
      }
      const double lim = L + sqrt(ha) + sqrt(hb);
      if (r2 >= lim * lim) continue;
      double sa, tb, d, n[3];
      ee_closest(a0, a1, b0, b1, sa, tb, d, n);
      if (d < L) {
        const int s = atomicAdd(&n_pairs, 1);
        if (s < kBallMaxPairs) plist[s] = (int)((2u << 30) | ((unsigned)pe << 15) | (unsigned)be_);
      }
    }
    __syncthreads();
    if (n_pairs > kBallMaxPairs) { if (tid == 0) { s_flags |= kBallFlagOverflow; n_pairs = kBallMaxPairs; } }
    __syncthreads();
    BALL_TICK(1);  // candidates, pair list
    // ---- nodal gradient, diagonal blocks (pad rows; pairs are added below), ground ----
    for (int v = tid; v < V; v += NT) {
      const double a3[3] = {vg[v * 3], vg[v * 3 + 1], vg[v * 3 + 2]};  // (the element pass's sums)
      const double mv = m.mass[v];
      const bool c = cons && cons[v];
      const double md = mv * (1.0 + (c ? m.strength : 0.0));
      double D[9] = {md, 0, 0, 0, md, 0, 0, 0, md};
      double gz = 0.0, cb = 0.0;
      const double w = bd.parea[v];
      if (bd.ground && w > 0.0) {
        double bb, b1, b2;
        barrier3((xs[v * 3 + 2] - bd.gh) / dhat, bb, b1, b2);
        if (!(xs[v * 3 + 2] - bd.gh > 0.0)) atomicOr(&s_flags, kFemFlagPenetration);
        gz = kk * w * b1 / dhat;
        cb = kk * w * b2 / (dhat * dhat);
        D[8] += cb;
        if (fric && take_lag && b1 < 0.0) {  // (inside the ground's barrier zone)
          const int sl = atomicAdd(&n_fric, 1);
          if (sl < kBallMaxFric) {
            double* rc = frec + (size_t)sl * kBallRec;
            rc[0] = -bd.kappa * w * b1 / dhat; rc[1] = 0.0; rc[2] = 0.0; rc[3] = 1.0;
            rc[4] = rc[5] = rc[6] = rc[7] = 0.0; rc[8] = 1.0; rc[9] = 0.0; rc[10] = 0.0;
            int* ri = reinterpret_cast<int*>(rc + 11);
            ri[0] = v; ri[1] = -1; ri[2] = -1;
          }
        }
      }
      cbp[v] = cb;
      if (blk) {  // elastic part: the step's lagged blocks (see blk above)
        const double* qb = blk + (size_t)v * 16;
        D[0] += qb[0]; D[1] += qb[1]; D[2] += qb[2]; D[3] += qb[1]; D[4] += qb[3]; D[5] += qb[4]; D[6] += qb[2]; D[7] += qb[4]; D[8] += qb[5];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) Dinv[(size_t)v * 9 + k] = D[k];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double gi = a3[i] + mv * (xs[v * 3 + i] - xt[v * 3 + i]);
        if (c) gi += m.strength * mv * (xs[v * 3 + i] - aim[v * 3 + i]);
        vg[v * 3 + i] = gi + (i == 2 ? gz : 0.0);
      }
    }
    if (bd.ground)
      for (int k = tid; k < nv; k += NT) {
        const double gap = xb[k * 3 + 2] - bd.gh;
        if (!(gap > 0.0)) atomicOr(&s_flags, kFemFlagPenetration);
        if (gap < dhat && gap > 0.0) {
          double bb, b1, b2;
          barrier3(gap / dhat, bb, b1, b2);
          const double f = kk * bd.area[k] * b1 / dhat, cb = kk * bd.area[k] * b2 / (dhat * dhat);
          if (fric && take_lag) {
            const int sl = atomicAdd(&n_fric, 1);
            if (sl < kBallMaxFric) {
              double* rc = frec + (size_t)sl * kBallRec;
              rc[0] = -bd.kappa * bd.area[k] * b1 / dhat; rc[1] = 0.0; rc[2] = 0.0; rc[3] = 1.0;
#pragma unroll
              for (int a4 = 0; a4 < 4; ++a4) rc[4 + a4] = bd.Y[k * 4 + a4];
              rc[8] = rc[9] = rc[10] = 0.0;
              int* ri = reinterpret_cast<int*>(rc + 11);
              ri[0] = -1; ri[1] = -1; ri[2] = -1;
            }
          }
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            atomicAdd(&gb[a * 3 + 2], f * bd.Y[k * 4 + a]);
#pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(&YY[a * 4 + c], cb * bd.Y[k * 4 + a] * bd.Y[k * 4 + c]);
          }
        }
      }
    __syncthreads();
    BALL_TICK(2);  // gradient, elastic blocks, ground
    // ---- pairs at x: gradient, diagonal blocks, ball block, active records ----
    SUB_TICK0();
    {
      const int np = n_pairs;
      for (int k = tid; k < np; k += NT) {
        const unsigned code = (unsigned)plist[k];
        const int kind = (int)(code >> 30), pi = (int)((code >> 15) & 0x7fff), tj = (int)(code & 0x7fff);
        double d, n[3], w, cq[4], pco[3];
        int prow[3];
        double molg = 0.0, ga[3] = {0.0, 0.0, 0.0}, gbq[3] = {0.0, 0.0, 0.0}, yd[4] = {0.0, 0.0, 0.0, 0.0};  // mollifier's own gradient (edge-edge pairs below the threshold)
        if (kind == 2) {
          const int* ea = bd.pedge + pi * 2;
          const int* eb = bd.bedge + tj * 2;
          double a0[3], a1[3], b0[3], b1[3], sa, tb, mol, dm, e1[3], e2[3], u[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) { a0[i] = xs[ea[0] * 3 + i]; a1[i] = xs[ea[1] * 3 + i]; b0[i] = xb[eb[0] * 3 + i]; b1[i] = xb[eb[1] * 3 + i]; }
          ee_closest(a0, a1, b0, b1, sa, tb, d, n);
          if (!(d < dhat)) continue;
          ee_mollifier(a0, a1, b0, b1, 1e-3 * bd.pelen2[pi] * bd.belen2[tj], mol, dm, e1, e2, u);
          const double w0 = 0.5 * (bd.pearea[pi] + bd.bearea[tj]);
          w = w0 * mol;
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) {
            cq[a4] = -((1.0 - tb) * bd.Y[eb[0] * 4 + a4] + tb * bd.Y[eb[1] * 4 + a4]);
            yd[a4] = bd.Y[eb[1] * 4 + a4] - bd.Y[eb[0] * 4 + a4];
          }
          prow[0] = ea[0]; prow[1] = ea[1]; prow[2] = -1;
          pco[0] = 1.0 - sa; pco[1] = sa; pco[2] = 0.0;
          if (dm > 0.0) {  // grad c = 2 (e2 x u) on a1 (minus on a0), 2 (u x e1) on b1 (minus on b0)
            molg = w0 * dm;
            ga[0] = 2.0 * (e2[1] * u[2] - e2[2] * u[1]); ga[1] = 2.0 * (e2[2] * u[0] - e2[0] * u[2]); ga[2] = 2.0 * (e2[0] * u[1] - e2[1] * u[0]);
            gbq[0] = 2.0 * (u[1] * e1[2] - u[2] * e1[1]); gbq[1] = 2.0 * (u[2] * e1[0] - u[0] * e1[2]); gbq[2] = 2.0 * (u[0] * e1[1] - u[1] * e1[0]);
          }
        } else {
          const int* tr = (kind == 0 ? bd.tri : bd.ptri) + tj * 3;
          const int t0 = tr[0], t1 = tr[1], t2 = tr[2];
          double p3[3], a[3], bq[3], c[3], be[3];
          if (kind == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { p3[i] = xs[pi * 3 + i]; a[i] = xb[t0 * 3 + i]; bq[i] = xb[t1 * 3 + i]; c[i] = xb[t2 * 3 + i]; }
            w = bd.parea[pi];
          } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) { p3[i] = xb[pi * 3 + i]; a[i] = xs[t0 * 3 + i]; bq[i] = xs[t1 * 3 + i]; c[i] = xs[t2 * 3 + i]; }
            w = bd.area[pi];
          }
          pt_closest(p3, a, bq, c, be, d, n);
          if (kind == 0) {
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) cq[a4] = -(be[0] * bd.Y[t0 * 4 + a4] + be[1] * bd.Y[t1 * 4 + a4] + be[2] * bd.Y[t2 * 4 + a4]);
            prow[0] = pi; prow[1] = -1; prow[2] = -1;
            pco[0] = 1.0; pco[1] = 0.0; pco[2] = 0.0;
          } else {
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) cq[a4] = bd.Y[pi * 4 + a4];
            prow[0] = t0; prow[1] = t1; prow[2] = t2;
            pco[0] = -be[0]; pco[1] = -be[1]; pco[2] = -be[2];
          }
        }
        if (!(d < dhat) || !(w > 0.0)) continue;
        double bb, b1, b2;
        barrier3(d / dhat, bb, b1, b2);
        const double s = kk * w * b1 / dhat, wk = kk * w * b2 / (dhat * dhat);
        if (molg > 0.0) {
          const double f = kk * molg * bb;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            atomicAdd(&vg[prow[0] * 3 + i], -f * ga[i]);
            atomicAdd(&vg[prow[1] * 3 + i], f * ga[i]);
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) atomicAdd(&gb[a4 * 3 + i], f * yd[a4] * gbq[i]);
          }
        }
        // pad rows
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (prow[r] < 0) continue;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            atomicAdd(&vg[prow[r] * 3 + i], s * pco[r] * n[i]);
#pragma unroll
            for (int j = 0; j < 3; ++j) atomicAdd(&Dinv[(size_t)prow[r] * 9 + i * 3 + j], wk * pco[r] * pco[r] * n[i] * n[j]);
          }
        }
        // ball rows
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            atomicAdd(&gb[a4 * 3 + i], s * cq[a4] * n[i]);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
              for (int j = 0; j < 3; ++j) atomicAdd(&Bm[(a4 * 3 + i) * 12 + c4 * 3 + j], wk * cq[a4] * cq[c4] * n[i] * n[j]);
          }
        if (take_lag) {
          const int sf = atomicAdd(&n_fric, 1);
          if (sf < kBallMaxFric) {
            double* rc = frec + (size_t)sf * kBallRec;
            rc[0] = -bd.kappa * w * b1 / dhat; rc[1] = n[0]; rc[2] = n[1]; rc[3] = n[2];
            rc[4] = cq[0]; rc[5] = cq[1]; rc[6] = cq[2]; rc[7] = cq[3];
            rc[8] = pco[0]; rc[9] = pco[1]; rc[10] = pco[2];
            int* ri = reinterpret_cast<int*>(rc + 11);
            ri[0] = prow[0]; ri[1] = prow[1]; ri[2] = prow[2];
          }
        }
        const int sl = atomicAdd(&n_act, 1);
        if (sl < kBallMaxActive) {
          double* rc = arec + (size_t)sl * kBallRec;
          rc[0] = wk; rc[1] = n[0]; rc[2] = n[1]; rc[3] = n[2];
          rc[4] = cq[0]; rc[5] = cq[1]; rc[6] = cq[2]; rc[7] = cq[3];
          rc[8] = pco[0]; rc[9] = pco[1]; rc[10] = pco[2];
          int* ri = reinterpret_cast<int*>(rc + 11);
          ri[0] = prow[0]; ri[1] = prow[1]; ri[2] = prow[2];
        }
      }
    }
    __syncthreads();
    if (n_act > kBallMaxActive) { if (tid == 0) { s_flags |= kBallFlagOverflow; n_act = kBallMaxActive; } }
    if (n_fric > kBallMaxFric) { if (tid == 0) { s_flags |= kBallFlagOverflow; n_fric = kBallMaxFric; } }
    SUB_TICK(0);
    if (fric) {  // friction of the lagged contacts at x: gradient, Hessians (kept for H.p), diagonal / ball blocks
      __syncthreads();
      const int nf = n_fric;
      // (records dealt to the waves round-robin: the ball rows of every record go to the SAME 12 + 144 LDS addresses, and lanes of one wave
      //  that add to one address are served one after the other - 60 records in wave 0 were 45 kcycles of this phase)
      for (int k = (tid & 63) * NWV + (tid >> 6); k < nf; k += NT) {
        const double* rc = frec + (size_t)k * kBallRec;
        const int* ri = reinterpret_cast<const int*>(rc + 11);
        double u[3];
        fric_u(rc, xs, qs, u);
        const double yv = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        const bool stick = yv < f_eps;
        const double fa = stick ? 2.0 / f_eps - yv / (f_eps * f_eps) : 1.0 / yv;
        const double fb = stick ? 2.0 / f_eps - 2.0 * yv / (f_eps * f_eps) : 0.0;
        const double cf = dt2 * f_mu * rc[0];
        const double iy = yv > 0.0 ? 1.0 / yv : 0.0;
        const double t3[3] = {u[0] * iy, u[1] * iy, u[2] * iy}, n3[3] = {rc[1], rc[2], rc[3]};
        double M[6];  // xx xy xz yy yz zz of cf [fa (I - n n^T) + (fb - fa) t t^T]
        {
          int kq = 0;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j) M[kq++] = cf * (fa * ((i == j ? 1.0 : 0.0) - n3[i] * n3[j]) + (fb - fa) * t3[i] * t3[j]);
        }
#pragma unroll
        for (int kq = 0; kq < 6; ++kq) fM[(size_t)k * 6 + kq] = M[kq];
        const double Mf[9] = {M[0], M[1], M[2], M[1], M[3], M[4], M[2], M[4], M[5]};
        const double g3[3] = {cf * fa * u[0], cf * fa * u[1], cf * fa * u[2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (ri[r] < 0) continue;
          const double pc = rc[8 + r];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            atomicAdd(&vg[ri[r] * 3 + i], pc * g3[i]);
#pragma unroll
            for (int j = 0; j < 3; ++j) atomicAdd(&Dinv[(size_t)ri[r] * 9 + i * 3 + j], pc * pc * Mf[i * 3 + j]);
          }
        }
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
          const double ca = rc[4 + a4];
          if (ca == 0.0) continue;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            atomicAdd(&gb[a4 * 3 + i], ca * g3[i]);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
              for (int j = 0; j < 3; ++j) atomicAdd(&Bm[(a4 * 3 + i) * 12 + c4 * 3 + j], ca * rc[4 + c4] * Mf[i * 3 + j]);
          }
        }
      }
      __syncthreads();
    }
    SUB_TICK(1);
    // ---- ball rows: gradient, the block without the pairs (H.p) and with them (preconditioner, factored) ----
    if (tid < 144) {
      const int ra = tid / 12, ca = tid - ra * 12, a4 = ra / 3, i = ra - a4 * 3, c4 = ca / 3, j = ca - c4 * 3;
      double v = (i == j) ? bd.S[a4 * 4 + c4] : 0.0;
      if (a4 > 0 && c4 > 0) {  // 4 kappa vol dt^2 [delta_mn A A^T + c_n c_m^T], m = a4 - 1, n = c4 - 1
        const int mm = a4 - 1, nn = c4 - 1;
        double blk = qs[3 + nn * 3 + i] * qs[3 + mm * 3 + j];
        if (mm == nn) blk += qs[3 + i] * qs[3 + j] + qs[6 + i] * qs[6 + j] + qs[9 + i] * qs[9 + j];
        v += dt2 * 4.0 * bd.kv * blk;
      }
      if (i == 2 && j == 2) v += YY[a4 * 4 + c4];
      B0[tid] = v;
      Lc[tid] = v + Bm[tid];
    }
    if (tid < 12) {
      const int a4 = tid / 3, i = tid - a4 * 3;
      double g = 0.0;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) g += bd.S[a4 * 4 + c4] * (qs[c4 * 3 + i] - qts[c4 * 3 + i]);
      if (a4 > 0) {
        const int mm = a4 - 1;
        double go = 0.0;
        for (int l = 0; l < 3; ++l) {
          const double r = qs[3 + mm * 3] * qs[3 + l * 3] + qs[4 + mm * 3] * qs[4 + l * 3] + qs[5 + mm * 3] * qs[5 + l * 3] - (mm == l ? 1.0 : 0.0);
          go += r * qs[3 + l * 3 + i];
        }
        g += dt2 * 4.0 * bd.kv * go;
      }
      vg[V * 3 + tid] = g + gb[tid];
    }
    __syncthreads();
    if (tid == 0) {  // Cholesky of the 12 x 12 block in place (lower triangle)
      for (int j = 0; j < 12; ++j) {
        double s = Lc[j * 12 + j];
        for (int k = 0; k < j; ++k) s -= Lc[j * 12 + k] * Lc[j * 12 + k];
        s = s > 0.0 ? sqrt(s) : sqrt(fmax(bd.S[0], 1e-300));  // (the block is SPD by construction: S (x) I is, the rest is PSD)
        Lc[j * 12 + j] = s;
        for (int i = j + 1; i < 12; ++i) {
          double t = Lc[i * 12 + j];
          for (int k = 0; k < j; ++k) t -= Lc[i * 12 + k] * Lc[j * 12 + k];
          Lc[i * 12 + j] = t / s;
        }
      }
    }
    __syncthreads();
    if (tid < 12) {  // column tid of the inverse: L L^T z = e_tid (twelve independent solves; applied by twelve dot products per PCG iteration)
      double yv[12];
      for (int i = 0; i < 12; ++i) {
        double sv = i == tid ? 1.0 : 0.0;
        for (int k = 0; k < i; ++k) sv -= Lc[i * 12 + k] * yv[k];
        yv[i] = sv / Lc[i * 12 + i];
      }
      for (int i = 11; i >= 0; --i) {
        double sv = yv[i];
        for (int k = i + 1; k < 12; ++k) sv -= Lc[k * 12 + i] * yv[k];
        yv[i] = sv / Lc[i * 12 + i];
      }
      for (int i = 0; i < 12; ++i) Bi[i * 12 + tid] = yv[i];
    }
    if (mode & 1) {  // terms only
      __syncthreads();
      const double E = energy(x, qs, xb);
      if (tid == 0 && e_out) e_out[b] = E;
      if (g_out)
        for (int k = tid; k < 3 * VN; k += NT) g_out[(size_t)b * 3 * VN + k] = vg[k];
      if (tid == 0 && step_info) step_info[(size_t)b * 4 + 2] = (double)s_flags;
      return;
    }
    SUB_TICK(2);
    // block part of the pad's preconditioner: block-tridiagonal LDL^T along the vertex chains of tacex_fem_set_chains (the columns of
    // vertices through the pad's thickness; a chain of one vertex = 3 x 3 block Jacobi) - S_0 = D_0, G_i = S_i^-1 E_i,
    // S_{i+1} = D_{i+1} - E_i^T G_i, the thread of a chain walks it (fem_newton_lds_kernel does the same); S^-1 | G as floats in LDS
    for (int ch = tid; ch < nch; ch += NT) {
      int v = m.ch_next ? m.ch_head[ch] : ch;
      double S[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) S[k] = Dinv[(size_t)v * 9 + k];
      while (true) {
        double Si[9];
        if (!inv3_spd(S, Si)) {
          const double dm = fmax(S[0], fmax(S[4], S[8]));
          const double im = 1.0 / (dm > 0.0 ? dm : 1.0);
          Si[0] = im; Si[1] = 0; Si[2] = 0; Si[3] = 0; Si[4] = im; Si[5] = 0; Si[6] = 0; Si[7] = 0; Si[8] = im;
        }
        float* f = cf + v * 15;
        f[0] = (float)Si[0]; f[1] = (float)Si[1]; f[2] = (float)Si[2]; f[3] = (float)Si[4]; f[4] = (float)Si[5]; f[5] = (float)Si[8];
        const int n = cnx[v] == 0xffff ? -1 : (int)cnx[v];
        if (n < 0 || !blk) {
#pragma unroll
          for (int k = 0; k < 9; ++k) f[6 + k] = 0.0f;
          if (n < 0) break;
#pragma unroll
          for (int k = 0; k < 9; ++k) S[k] = Dinv[(size_t)n * 9 + k];
          v = n;
          continue;
        }
        const double* Ev = blk + (size_t)v * 16 + 6;
        double G[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) G[i * 3 + k] = Si[i * 3 + 0] * Ev[k] + Si[i * 3 + 1] * Ev[3 + k] + Si[i * 3 + 2] * Ev[6 + k];
#pragma unroll
        for (int k = 0; k < 9; ++k) f[6 + k] = (float)G[k];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k)
            S[i * 3 + k] = Dinv[(size_t)n * 9 + i * 3 + k] - (Ev[i] * G[k] + Ev[3 + i] * G[3 + k] + Ev[6 + i] * G[6 + k]);
        v = n;
      }
    }
    __syncthreads();  // (the chains have read their blocks: r may take over the tail of their slot)
    for (int k = tid; k < 3 * VN; k += NT) { rL[k] = (bd.kinematic && k >= 3 * V) ? 0.0 : -vg[k]; dL[k] = 0.0; }  // (a fixed body: zero residual rows stay zero through the PCG)
    __syncthreads();
    SUB_TICK(3);
    BALL_TICK(3);  // pairs, friction, ball blocks, factorisation, pad block inverses
    // ---- PCG: every vector in LDS (x, p, H.p = the accumulators of the tets' rows, z, r, d); the mesh constants, the blocks' tables and
    //      the pair records are what it reads from memory ----
    for (int k = tid; k < 3 * V; k += NT) acc[k] = 0.0;  // (xs holds x since the kernel's start / the last accepted step)
    __syncthreads();
    constexpr int VS = 1024 / NT;  // vertices per thread (V <= 780: ball_lds_ok)
    double md_r[VS], cb_r[VS];  // this thread's vertices: mass (+ constraint) diagonal and ground curvature, constant through the PCG loop
#pragma unroll
    for (int sl = 0; sl < VS; ++sl) {
      const int v = tid + sl * NT;
      md_r[sl] = v < V ? m.mass[v] * (1.0 + ((cons && cons[v]) ? m.strength : 0.0)) : 0.0;
      cb_r[sl] = v < V ? cbp[v] : 0.0;
    }
    double rz = precondition();
    for (int k = tid; k < 3 * V; k += NT) ps[k] = rsl[k];
    if (tid < 12) ps[V * 3 + tid] = zq[tid];
    const double rz0 = rz;
    int it = 0;
    __syncthreads();
    while (it < pcg_max_iter && rz0 > 0.0 && rz > pcg_tol_rate * rz0) {
      PCG_TICK0();
      for (int t = tid; t < T; t += NT) {
        int v[4];
        double Di[9], F[9], dF[9], dP[9], r[12];
        load_tet(m, t, v, Di);
        deformation_gradient(xs, v, Di, F);
        TetState s;
        tet_state(m, F, s);
        deformation_gradient(ps, v, Di, dF);
        apply_dP(m, s, dF, dP);
        shape_rows(Di, r);
        const double sc = dt2 * m.vol[t];
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4)
#pragma unroll
          for (int i = 0; i < 3; ++i)
            atomicAdd(&acc[v[w4] * 3 + i], sc * (dP[i * 3 + 0] * r[w4 * 3 + 0] + dP[i * 3 + 1] * r[w4 * 3 + 1] + dP[i * 3 + 2] * r[w4 * 3 + 2]));
      }
      __syncthreads();
      PCG_TICK(0);
#pragma unroll
      for (int sl = 0; sl < VS; ++sl) {
        const int v = tid + sl * NT;
        if (v < V) {
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[v * 3 + i] += md_r[sl] * ps[v * 3 + i] + (i == 2 ? cb_r[sl] * ps[v * 3 + 2] : 0.0);  // acc IS H.p from here on
        }
      }
      if (tid < 12) {
        double sv = 0.0;
        for (int k = 0; k < 12; ++k) sv += B0[tid * 12 + k] * ps[V * 3 + k];
        Hpq[tid] = sv;
      }
      if (tid >= 64 && tid < 64 + NWV * 12) Hw[tid - 64] = 0.0;
      __syncthreads();
      PCG_TICK(6);
      // records are dealt to the waves round-robin (record k -> wave k % 8, lane k / 8) and each wave adds the ball rows into its own copy
      double* const hw = Hw + 12 * (tid >> 6);
      {
        const int na = n_act;
        for (int k = (tid & 63) * NWV + (tid >> 6); k < na; k += NT) {
          const double* rc = arec + (size_t)k * kBallRec;
          const int* ri = reinterpret_cast<const int*>(rc + 11);
          const double n0 = rc[1], n1 = rc[2], n2 = rc[3];
          double gp = 0.0;
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) gp += rc[4 + a4] * (n0 * ps[(V + a4) * 3] + n1 * ps[(V + a4) * 3 + 1] + n2 * ps[(V + a4) * 3 + 2]);
#pragma unroll
          for (int r = 0; r < 3; ++r)
            if (ri[r] >= 0) gp += rc[8 + r] * (n0 * ps[ri[r] * 3] + n1 * ps[ri[r] * 3 + 1] + n2 * ps[ri[r] * 3 + 2]);
          const double f = rc[0] * gp;
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4) {
            atomicAdd(&hw[a4 * 3], f * rc[4 + a4] * n0);
            atomicAdd(&hw[a4 * 3 + 1], f * rc[4 + a4] * n1);
            atomicAdd(&hw[a4 * 3 + 2], f * rc[4 + a4] * n2);
          }
#pragma unroll
          for (int r = 0; r < 3; ++r)
            if (ri[r] >= 0) {
              atomicAdd(&acc[ri[r] * 3], f * rc[8 + r] * n0);
              atomicAdd(&acc[ri[r] * 3 + 1], f * rc[8 + r] * n1);
              atomicAdd(&acc[ri[r] * 3 + 2], f * rc[8 + r] * n2);
            }
        }
      }
      if (fric) {
        const int nf = n_fric;
        for (int k = (tid & 63) * NWV + (tid >> 6); k < nf; k += NT) {
          const double* rc = frec + (size_t)k * kBallRec;
          const int* ri = reinterpret_cast<const int*>(rc + 11);
          const double* M = fM + (size_t)k * 6;
          double w3[3] = {0.0, 0.0, 0.0};
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
            for (int i = 0; i < 3; ++i) w3[i] += rc[4 + a4] * ps[(V + a4) * 3 + i];
#pragma unroll
          for (int r = 0; r < 3; ++r)
            if (ri[r] >= 0)
#pragma unroll
              for (int i = 0; i < 3; ++i) w3[i] += rc[8 + r] * ps[ri[r] * 3 + i];
          const double mw[3] = {M[0] * w3[0] + M[1] * w3[1] + M[2] * w3[2], M[1] * w3[0] + M[3] * w3[1] + M[4] * w3[2],
                                M[2] * w3[0] + M[4] * w3[1] + M[5] * w3[2]};
#pragma unroll
          for (int a4 = 0; a4 < 4; ++a4)
            if (rc[4 + a4] != 0.0)
#pragma unroll
              for (int i = 0; i < 3; ++i) atomicAdd(&hw[a4 * 3 + i], rc[4 + a4] * mw[i]);
#pragma unroll
          for (int r = 0; r < 3; ++r)
            if (ri[r] >= 0)
#pragma unroll
              for (int i = 0; i < 3; ++i) atomicAdd(&acc[ri[r] * 3 + i], rc[8 + r] * mw[i]);
        }
      }
      __syncthreads();
      PCG_TICK(1);
      double part = 0.0;
      for (int k = tid; k < 3 * V; k += NT) part += ps[k] * acc[k];
      if (tid < 12) {
#pragma unroll
        for (int w8 = 0; w8 < NWV; ++w8) Hpq[tid] += Hw[w8 * 12 + tid];
        if (bd.kinematic) Hpq[tid] = 0.0;  // a fixed body: its rows of the operator are eliminated (only thread tid reads Hpq[tid] below)
        part += ps[V * 3 + tid] * Hpq[tid];
      }
      const double pHp = bsum(part);
      if (!(pHp > 0.0)) {
        if (it == 0) {
          for (int k = tid; k < 3 * V; k += NT) dL[k] = rsl[k];
          if (tid < 12) dL[V * 3 + tid] = zq[tid];
        }
        break;
      }
      const double al = rz / pHp;
      for (int k = tid; k < 3 * V; k += NT) { dL[k] += al * ps[k]; rL[k] -= al * acc[k]; acc[k] = 0.0; }  // (acc: zero for the next sweep)
      if (tid < 12) { dL[V * 3 + tid] += al * ps[V * 3 + tid]; rL[V * 3 + tid] -= al * Hpq[tid]; }
      __syncthreads();
      PCG_TICK(2);
      const double rz_new = precondition();
      const double beta = rz_new / rz;
      for (int k = tid; k < 3 * V; k += NT) ps[k] = rsl[k] + beta * ps[k];
      if (tid < 12) ps[V * 3 + tid] = zq[tid] + beta * ps[V * 3 + tid];
      rz = rz_new;
      ++it;
      __syncthreads();
    }
    __syncthreads();
    pcg_total += it;
    BALL_TICK(4);  // PCG
    // ---- step bound ----
    if (tid < 12) rhs12[tid] = dL[V * 3 + tid];
    __syncthreads();
    ball_points(rhs12, dxb);
    __syncthreads();
    double amax = 1.0, vmax = 0.0, dmx = 0.0, dmc = 0.0;
    for (int v = tid; v < V; v += NT) {
      const double d0 = dL[v * 3], d1 = dL[v * 3 + 1], d2 = dL[v * 3 + 2];
      dmx = fmax(dmx, fmax(fabs(d0), fmax(fabs(d1), fabs(d2))));
      if (bd.parea[v] > 0.0) {
        vmax = fmax(vmax, sqrt(d0 * d0 + d1 * d1 + d2 * d2));
        const double gap = xs[v * 3 + 2] - bd.gh;
        if (bd.ground && d2 < 0.0 && gap > 0.0) amax = fmin(amax, kCcdSlack * gap / -d2);
      }
    }
    for (int k = tid; k < nv; k += NT) {
      const double d0 = dxb[k * 3], d1 = dxb[k * 3 + 1], d2 = dxb[k * 3 + 2];
      vmax = fmax(vmax, sqrt(d0 * d0 + d1 * d1 + d2 * d2));
      const double gap = xb[k * 3 + 2] - bd.gh;
      if (bd.ground && d2 < 0.0 && gap > 0.0) amax = fmin(amax, kCcdSlack * gap / -d2);
    }
    if (tid < 3) dmx = fmax(dmx, fabs(dL[V * 3 + tid]));
    if (tid >= 3 && tid < 12) dmc = fabs(dL[V * 3 + tid]);
    vmax = block_sum_max(vmax, sh);
    dmx = block_sum_max(dmx, sh);
    dmc = block_sum_max(dmc, sh);
    if (vmax > 0.0) amax = fmin(amax, kCcdSlack * R / (2.0 * vmax));
    amax = -block_sum_max(-amax, sh);
    {
      double al = amax;
      const int np = n_pairs;
      for (int k = tid; k < np; k += NT) {
        const unsigned code = (unsigned)plist[k];
        const int kind = (int)(code >> 30), pi = (int)((code >> 15) & 0x7fff), tj = (int)(code & 0x7fff);
        if (kind == 2) {
          const int* ea = bd.pedge + pi * 2;
          const int* eb = bd.bedge + tj * 2;
          double pa[6], pb[6], da[6], db[6];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              pa[j * 3 + i] = xs[ea[j] * 3 + i]; da[j * 3 + i] = dL[ea[j] * 3 + i];
              pb[j * 3 + i] = xb[eb[j] * 3 + i]; db[j * 3 + i] = dxb[eb[j] * 3 + i];
            }
          double sa, tb, d, n[3];
          ee_closest(pa, pa + 3, pb, pb + 3, sa, tb, d, n);
          if (d < R) al = fmin(al, accd_ee(pa, pb, da, db, al));
          continue;
        }
        const int* tr = (kind == 0 ? bd.tri : bd.ptri) + tj * 3;
        double p3[3], trx[9], dp[3], dtr[9];
        const double* P = kind == 0 ? xs : xb;
        const double* dP_ = kind == 0 ? dL : dxb;
        const double* Tq = kind == 0 ? xb : xs;
        const double* dT = kind == 0 ? dxb : dL;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          p3[i] = P[pi * 3 + i]; dp[i] = dP_[pi * 3 + i];
#pragma unroll
          for (int j = 0; j < 3; ++j) { trx[j * 3 + i] = Tq[tr[j] * 3 + i]; dtr[j * 3 + i] = dT[tr[j] * 3 + i]; }
        }
        double be[3], d, n[3];
        pt_closest(p3, trx, trx + 3, trx + 6, be, d, n);
        if (d < R) al = fmin(al, accd_pt(p3, trx, dp, dtr, al));
      }
      amax = -block_sum_max(-al, sh);
    }
    dmax_x = dmx; dmax_c = dmc;
    // ---- backtracking line search (first E <= E0 wins; rescue halvings as in the other Newton kernels) ----
    BALL_TICK(5);  // step bound (ground, additive CCD)
    // E(x): the accepted candidate's energy of the previous iteration IS this iteration's (every pair inside d_hat is in both lists)
    const double E0 = have_e ? e_carry : energy(xs, qs, xb);
    double step = amax, E1 = E0;
    bool accepted = false;
    const int ls_cap = ls_max_iter > kLsRescueStream ? ls_max_iter : kLsRescueStream;
    for (int ls = 0; ls <= ls_cap; ++ls) {
      for (int k = tid; k < 3 * V; k += NT) ycl[k] = xs[k] + step * dL[k];  // (the candidate in the idle PCG direction's slot; dL still holds d)
      if (tid < 12) rhs12[tid] = qs[tid] + step * dL[V * 3 + tid];
      __syncthreads();
      ball_points(rhs12, xbc);
      __syncthreads();
      const double Ec = energy(ycl, rhs12, xbc);
      if (Ec <= E0) { E1 = Ec; accepted = true; e_carry = Ec; have_e = true; break; }
      step *= 0.5;
      __syncthreads();
    }
    // REFINEMENT (mode bits 8-11 = bisections: tacex_fem_set_line_search_refine): a step the halving had to cut was cut by a pair ENTERING the barrier zone (10 GPa against a
    // 0.1 MPa gel: a few um inside cost more than the step gains, r06 section 15), and the accepted half usually leaves that pair just
    // OUTSIDE - where it has no curvature for the next iteration either, which is then cut again.  A few bisections between the accepted and
    // the last rejected step find a larger one that still decreases E and has the pair inside, active in the next Hessian.
    const int n_refine = (mode >> 8) & 15;
    if (accepted && step < amax && n_refine > 0) {
      double lo_s = step, hi_s = 2.0 * step;
      for (int rf = 0; rf < n_refine; ++rf) {
        const double mid = 0.5 * (lo_s + hi_s);
        __syncthreads();
        for (int k = tid; k < 3 * V; k += NT) ycl[k] = xs[k] + mid * dL[k];
        if (tid < 12) rhs12[tid] = qs[tid] + mid * dL[V * 3 + tid];
        __syncthreads();
        ball_points(rhs12, xbc);
        __syncthreads();
        const double Ec = energy(ycl, rhs12, xbc);
        if (Ec <= E0) { lo_s = mid; E1 = Ec; e_carry = Ec; } else { hi_s = mid; }
      }
      // (the candidate buffers hold the LAST trial: rebuild them at the accepted step)
      step = lo_s;
      __syncthreads();
      for (int k = tid; k < 3 * V; k += NT) ycl[k] = xs[k] + step * dL[k];
      if (tid < 12) rhs12[tid] = qs[tid] + step * dL[V * 3 + tid];
      __syncthreads();
      ball_points(rhs12, xbc);
      __syncthreads();
    }
    ++n_newton;
    BALL_TICK(6);  // line search
#ifdef TACEX_BALL_TRACE  // debug build: the late Newton iterations of the envs that need them (scripts/r06/ball_trace.py)
    if (tid == 0 && (nit >= 2 || it >= 20))
      printf("trace env %d newton %d: pcg %d amax %.3e step %.3e E0 %.12e dE %.3e dmx %.3e (tol %.1e) dmc %.3e (tol %.1e) pairs %d act %d fric %d\n", b, nit, it, amax, step,
             E0, E1 - E0, dmx, dx_tol, dmc, dc_tol, n_pairs, n_act, n_fric);
#endif
    if (accepted) {
      for (int k = tid; k < 3 * V; k += NT) { const double v_ = ycl[k]; x[k] = v_; xs[k] = v_; }
      for (int k = tid; k < 3 * nv; k += NT) xb[k] = xbc[k];
      __syncthreads();
      if (tid < 12) qs[tid] = rhs12[tid];
    } else if (!(dmx <= dx_tol && dmc <= dc_tol)) {
      if (tid == 0) s_flags |= kFemFlagLsFailed;
      __syncthreads();
      break;
    }
    __syncthreads();
    if (dmx <= dx_tol && dmc <= dc_tol) break;
  }
#ifdef TACEX_BALL_CLOCK
  if (tid == 0 && (b == 0 || b == (int)gridDim.x - 1))
    printf("ball clock env %d: newton %d pcg %d | kcycles: element %lld candidates %lld gradient+blocks %lld pairs+factor %lld PCG %lld stepbound %lld linesearch %lld | inside PCG: sweep %lld vertex %lld records %lld dot+update %lld restrict+chains %lld coarse %lld prolong+sum %lld | phase 3: pairs %lld friction %lld ball block %lld chains %lld\n", b,
           n_newton, pcg_total, bck[0] / 1000, bck[1] / 1000, bck[2] / 1000, bck[3] / 1000, bck[4] / 1000, bck[5] / 1000, bck[6] / 1000, pck[0] / 1000, pck[6] / 1000, pck[1] / 1000,
           pck[2] / 1000, pck[3] / 1000, pck[4] / 1000, pck[5] / 1000, sck[0] / 1000, sck[1] / 1000, sck[2] / 1000, sck[3] / 1000);
#endif
  if (tid < 12) q[tid] = qs[tid];
  if (tid == 0 && step_info) {
    double* si = step_info + (size_t)b * 4;
    si[0] = (double)n_newton; si[1] = fmax(dmax_x, dmax_c); si[2] = (double)s_flags; si[3] = (double)pcg_total;
  }
}

// predictor / velocity of the ball rows (the pad's go through fem_predict_kernel / fem_velocity_kernel)
__global__ __launch_bounds__(256) void fem_ball_predict_kernel(const double* __restrict__ q, const double* __restrict__ qv, double* __restrict__ qt,
                                                               double* __restrict__ qprev, int B, double dt, double g0, double g1, double g2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= B * 12) return;
  const int r = k % 12;
  const double g = r == 0 ? g0 : (r == 1 ? g1 : (r == 2 ? g2 : 0.0));  // gravity acts on the translation row p
  qprev[k] = q[k];
  qt[k] = q[k] + dt * qv[k] + dt * dt * g;
}
__global__ __launch_bounds__(256) void fem_ball_velocity_kernel(const double* __restrict__ q, const double* __restrict__ qprev, double* __restrict__ qv,
                                                                int B, double inv_dt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < B * 12) qv[k] = (q[k] - qprev[k]) * inv_dt;
}
