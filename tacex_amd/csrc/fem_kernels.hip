// Gelpad FEM inner step for MI355X (gfx950), float64, batched over independent environments.
//
// What libuipc's world.advance() executes for the TacEx gelpad (tacex_uipc/sim/uipc_sim.py:250-252;
// StableNeoHookean + ElasticModuli.youngs_poisson, tacex_uipc/objects/uipc_object.py:442-470; soft position
// constraints, tacex_uipc/sim/uipc_attachments.py:139-142,364-428).  libuipc is an un-vendored submodule, so
// this follows the published model (Smith, de Goes, Kim 2018, eq. 14) - see oracle/fem_oracle.py, PARITY UNPINNED.
//
//   Psi(F) = mu/2 (Ic-3) + lam/2 (J-alpha)^2 - mu/2 log(Ic+1) - Psi(I)
//   P      = a F + c C,         a = mu (1 - 1/(Ic+1)),  c = lam (J - alpha),  C = cof F = [f1xf2, f2xf0, f0xf1]
//   dP[dF] = a dF + b (F:dF) F + lam (C:dF) C + c dC[dF],   b = 2 mu / (Ic+1)^2
//   E(x)   = 1/2 sum m |x-xt|^2 + dt^2 sum vol Psi(F) + 1/2 s sum_{constrained} m |x-aim|^2
//
// Design (bandwidth-bound, no MFMA): one tet per lane for element terms with element-minor SoA outputs
// (energy (B,T), grad (B,12,T), hess (B,144,T)) so every store is coalesced; nodal assembly is an atomics-free
// vertex gather over a CSR incidence list; reductions are wave shuffles + LDS; the Newton step runs as ONE
// workgroup per environment - matrix-free PCG (block-Jacobi preconditioned) and the backtracking line search
// stay inside a single launch with no host round trip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <utility>
#include <vector>

#include "tacex_hip.h"
#include "tacex_internal.h"

namespace tacex {

struct FemDev {
  int V, T;
  const int* tets;        // (4,T) SoA
  const double* dminv;    // (9,T) SoA, row-major 3x3 per tet
  const double* vol;      // (T)
  const double* tet_rec;  // (T,12) AoS copy of one tet: vertex ids (4 ints in the first two doubles) | dminv (9) | vol - for loops that visit
                          // tets in VERTEX order (every lane another tet): six 16-byte loads per tet instead of 14 scattered ones; nullable
  const double* tet_blk;  // wave-blocked SoA copy for loops that visit tets in TET order (the sweeps of fem_newton_lds_kernel): per block of
                          // 64 tets [9][64] dminv | [64] vol | [4][64] vertex ids (int32) = kTetBlkBytes; every component of lane l sits at
                          // block base + immediate + 8 l, so ONE 32-bit offset register addresses all 14 loads (the SoA arrays above
                          // need 14 per-lane 64-bit addresses, which the Newton kernel spilled and re-read from scratch one by one)
  const double* mass;     // (V)
  const int* vt_off;      // (V+1) CSR vertex -> incident (tet*4 + local)
  const int* vt_idx;
  double mu, lam, alpha, psi_rest, dt, strength;
  double step_cap;  // bounding-box diagonal of the rest mesh: no line search starts with a vertex moving further (fem_newton_lds_kernel)
  // IPC contact of the gelpad surface against one analytic indenter per env (first slice)
  const double* area;       // (V) contact weight of a vertex = a third of the area of its surface triangles (0: interior); nullable
  const double* indenters;  // (B,8) [kind, cx, cy, cz, radius, nx, ny, nz]: kind 0 none, 1 sphere, 2 half-space; nullable
  double dhat, kappa;       // barrier activation distance [m], stiffness [J/m^2]
  double fric_mu, fric_eps; // Coulomb friction ratio (0: off) and stick tolerance eps_velocity * dt [m] (tacex_fem_set_friction)
  // coarse space of the two-level preconditioner (tacex_fem_set_coarse_space); nc = 0: block Jacobi alone
  int nc;                   // coarse nodes (<= kFemMaxCoarse)
  const int* cv_node;       // (V,8) coarse nodes of a vertex (trilinear hats of a coarse grid over the mesh)
  const double* cv_w;       // (V,8) their weights
  const int* cn_off;        // (nc+1) CSR coarse node -> (vertex, weight) of its support
  const int* cn_vtx;
  const double* cn_w;
  const double* ac_inv;     // (3 nc, 3 nc) inverse of P^T A_0 P, A_0 = rest-state operator incl. the constraint masses
  // rigid triangle-mesh indenter shared by all envs (indenter kind 4, tacex_fem_set_indenter_mesh)
  int im_nt;                // triangles
  const double* im_tri;     // (nt,9) a | b - a | c - a in the mesh frame
  const double* im_bs;      // (nt,4) bounding sphere: centroid, radius
  const double* im_cl;      // (ceil(nt / 16),4) bounding sphere of every cluster of 16 consecutive triangles (Morton order)
  // vertex chains of the block-tridiagonal part of the preconditioner (tacex_fem_set_chains); nullptr: every vertex its own chain
  int nch;                  // chains, singletons included (<= V)
  const int* ch_head;       // (nch) first vertex of every chain
  const int* ch_next;       // (V) successor in the chain, -1 at its end
  const int* ch_prev;       // (V) predecessor, -1 at its head
};
constexpr int kFemMaxCoarse = 64;

// ---- IPC barrier of one surface vertex against the env's analytic indenter ------------------------------------------
// Li et al. 2020 (IPC) eq. 6 in the dimensionless gap s = d / dhat:  b(s) = -(s - 1)^2 ln s  for 0 < s < 1, 0 beyond.
// Potential term of a vertex with weight w: dt^2 kappa w b(d / dhat); d = signed distance to the indenter surface
// (sphere: |x - c| - R, half-space: n . (x - c), capsule: distance to the axis segment - R), n = grad d.  A gap <= 0 is a penetration: infinite energy (the
// line search never accepts it; the conservative step bound below keeps the Newton direction out of it).
typedef double v4d __attribute__((ext_vector_type(4)));
struct ContactEval {
  bool active;      // 0 < d < dhat
  bool penetrating; // d <= 0
  double d, n[3];
  double e, b1, b2; // energy, dE/dd, d2E/dd2 (already times kappa w, NOT times dt^2)
};
// Unsigned distance of p (mesh frame) to the nearest triangle of the indenter mesh and the unit vector from the closest point to p.
// Two-level culling with bounding spheres: clusters of kMeshCluster triangles (Morton order of the centroids, built on the host), then
// the triangles of a cluster; a sphere farther than the best distance so far is skipped.  The sphere tables are fetched FOUR at a time
// (a loop with one dependent L2 round trip per triangle took 45 ms per step for 320 triangles).  `cut2`: the search radius squared -
// energy evaluations only need triangles within d_hat (+ offset); with nothing inside the result is sqrt(cut2), n = 0.  Closest point
// by Ericson (Real-Time Collision Detection 5.1.5), regions in the book's order; of two triangles at exactly the same distance the
// first visited wins (they share the closest point unless p lies on the medial axis).
constexpr int kMeshCluster = 16;
struct MeshDist { double d, n0, n1, n2; };
__device__ __noinline__ MeshDist mesh_distance(int nt, const double* __restrict__ tris, const double* __restrict__ bsph,
                                               const double* __restrict__ clus, double p0, double p1, double p2, double cut2) {
  const double p[3] = {p0, p1, p2};
  double best2 = cut2, best = sqrt(cut2), bq[3] = {0, 0, 0};
  auto beyond = [&](const v4d& sp) {  // the sphere (centre, radius) lies farther than the best distance so far
    const double c0 = p[0] - sp.x, c1 = p[1] - sp.y, c2 = p[2] - sp.z;
    const double lim = sp.w + best;
    return c0 * c0 + c1 * c1 + c2 * c2 >= lim * lim;
  };
  auto triangle = [&](int t) {
    const double* tr = tris + (size_t)t * 9;
    const double a[3] = {tr[0], tr[1], tr[2]}, ab[3] = {tr[3], tr[4], tr[5]}, ac[3] = {tr[6], tr[7], tr[8]};
    const double ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
    const double d1 = ab[0] * ap[0] + ab[1] * ap[1] + ab[2] * ap[2], d2 = ac[0] * ap[0] + ac[1] * ap[1] + ac[2] * ap[2];
    const double bp[3] = {ap[0] - ab[0], ap[1] - ab[1], ap[2] - ab[2]};
    const double d3 = ab[0] * bp[0] + ab[1] * bp[1] + ab[2] * bp[2], d4 = ac[0] * bp[0] + ac[1] * bp[1] + ac[2] * bp[2];
    const double cp[3] = {ap[0] - ac[0], ap[1] - ac[1], ap[2] - ac[2]};
    const double d5 = ab[0] * cp[0] + ab[1] * cp[1] + ab[2] * cp[2], d6 = ac[0] * cp[0] + ac[1] * cp[1] + ac[2] * cp[2];
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    double s = 0.0, u = 0.0;  // closest point = a + s ab + u ac
    if (d1 <= 0.0 && d2 <= 0.0) { s = 0.0; u = 0.0; }
    else if (d3 >= 0.0 && d4 <= d3) { s = 1.0; u = 0.0; }
    else if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { s = d1 / (d1 - d3); u = 0.0; }
    else if (d6 >= 0.0 && d5 <= d6) { s = 0.0; u = 1.0; }
    else if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { s = 0.0; u = d2 / (d2 - d6); }
    else if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) { u = (d4 - d3) / ((d4 - d3) + (d5 - d6)); s = 1.0 - u; }
    else { const double den = 1.0 / (va + vb + vc); s = vb * den; u = vc * den; }
    const double q[3] = {a[0] + s * ab[0] + u * ac[0], a[1] + s * ab[1] + u * ac[1], a[2] + s * ab[2] + u * ac[2]};
    const double r0 = p[0] - q[0], r1 = p[1] - q[1], r2 = p[2] - q[2];
    const double dd = r0 * r0 + r1 * r1 + r2 * r2;
    if (dd < best2) { best2 = dd; best = sqrt(dd); bq[0] = r0; bq[1] = r1; bq[2] = r2; }
  };
  const v4d* cl4 = reinterpret_cast<const v4d*>(clus);
  const v4d* bs4 = reinterpret_cast<const v4d*>(bsph);
  const int ncl = (nt + kMeshCluster - 1) / kMeshCluster;
  auto cluster = [&](int q) {
    const int t0 = q * kMeshCluster, t1 = min(nt, t0 + kMeshCluster);
    for (int tb = t0; tb < t1; tb += 4) {
      v4d ts[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ts[j] = bs4[min(tb + j, nt - 1)];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (tb + j < t1 && !beyond(ts[j])) triangle(tb + j);
    }
  };
  // pass 1: the cluster whose sphere comes nearest is searched first - its best distance culls nearly all of pass 2 (walking the
  // clusters in table order the bound only tightens as fast as the order happens to approach p)
  int first = -1;
  {
    double lo = 1e300;
    for (int c0 = 0; c0 < ncl; c0 += 4) {
      v4d cs[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) cs[k] = cl4[min(c0 + k, ncl - 1)];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double c0x = p[0] - cs[k].x, c1x = p[1] - cs[k].y, c2x = p[2] - cs[k].z;
        const double lb = sqrt(c0x * c0x + c1x * c1x + c2x * c2x) - cs[k].w;
        if (c0 + k < ncl && lb < lo) { lo = lb; first = c0 + k; }
      }
    }
    if (first >= 0 && lo < best) cluster(first);
  }
  for (int c0 = 0; c0 < ncl; c0 += 4) {
    v4d cs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cs[k] = cl4[min(c0 + k, ncl - 1)];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c0 + k < ncl && c0 + k != first && !beyond(cs[k])) cluster(c0 + k);
  }
  MeshDist r;
  r.d = best;
  const double ir = best > 0.0 && best2 < cut2 ? 1.0 / best : 0.0;
  r.n0 = bq[0] * ir; r.n1 = bq[1] * ir; r.n2 = bq[2] * ir;
  return r;
}

// MESH = false compiles the triangle-mesh indenter (kind 4: a function call in the middle of a 256-register kernel) out: the Newton
// kernel is instantiated both ways and the mesh-capable one is launched only when a mesh has been set.
template <bool MESH = true>
__device__ __forceinline__ ContactEval contact_eval(const FemDev& m, const double* ind, double w, const double x[3], bool need_distance = true) {
  ContactEval c;
  c.active = false; c.penetrating = false; c.d = 1e300; c.e = 0.0; c.b1 = 0.0; c.b2 = 0.0; c.n[0] = c.n[1] = c.n[2] = 0.0;
  if (!ind || !(w > 0.0)) return c;
  const int kind = (int)ind[0];
  if (kind == 1) {
    const double r0 = x[0] - ind[1], r1 = x[1] - ind[2], r2 = x[2] - ind[3];
    const double rho = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
    c.d = rho - ind[4];
    const double ir = rho > 0.0 ? 1.0 / rho : 0.0;
    c.n[0] = r0 * ir; c.n[1] = r1 * ir; c.n[2] = r2 * ir;
  } else if (kind == 2) {
    c.n[0] = ind[5]; c.n[1] = ind[6]; c.n[2] = ind[7];
    c.d = c.n[0] * (x[0] - ind[1]) + c.n[1] * (x[1] - ind[2]) + c.n[2] * (x[2] - ind[3]);
  } else if (kind == 3) {
    // capsule (cylinder with hemispherical caps, e.g. a lying pin or a finger): centre c, radius R, the vector (nx, ny, nz) is
    // HALF the axis (direction and half length); the closest axis point is c + clamp(p . a / |a|^2, -1, 1) a
    const double p0 = x[0] - ind[1], p1 = x[1] - ind[2], p2 = x[2] - ind[3];
    const double a0 = ind[5], a1 = ind[6], a2 = ind[7];
    const double aa = a0 * a0 + a1 * a1 + a2 * a2;
    double t = aa > 0.0 ? (p0 * a0 + p1 * a1 + p2 * a2) / aa : 0.0;
    t = t < -1.0 ? -1.0 : (t > 1.0 ? 1.0 : t);
    const double r0 = p0 - t * a0, r1 = p1 - t * a1, r2 = p2 - t * a2;
    const double rho = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
    c.d = rho - ind[4];
    const double ir = rho > 0.0 ? 1.0 / rho : 0.0;
    c.n[0] = r0 * ir; c.n[1] = r1 * ir; c.n[2] = r2 * ir;
  } else if (MESH && kind == 4 && m.im_nt > 0) {
    // rigid triangle mesh (tacex_fem_set_indenter_mesh) at position c with rotation vector (nx, ny, nz), inflated by R: UNSIGNED
    // distance to the nearest triangle - R (the step bound keeps a vertex from crossing the surface; a vertex that starts
    // inside the mesh is not detected)
    const double r0 = ind[5], r1 = ind[6], r2 = ind[7];
    const double th2 = r0 * r0 + r1 * r1 + r2 * r2, th = sqrt(th2);
    const double ka = th < 1e-12 ? 1.0 : sin(th) / th, kb = th < 1e-12 ? 0.0 : (1.0 - cos(th)) / th2;
    // R = I + ka K + kb K^2, K = [r]x
    const double R[9] = {1.0 - kb * (r1 * r1 + r2 * r2), -ka * r2 + kb * r0 * r1, ka * r1 + kb * r0 * r2,
                         ka * r2 + kb * r0 * r1, 1.0 - kb * (r0 * r0 + r2 * r2), -ka * r0 + kb * r1 * r2,
                         -ka * r1 + kb * r0 * r2, ka * r0 + kb * r1 * r2, 1.0 - kb * (r0 * r0 + r1 * r1)};
    const double g0 = x[0] - ind[1], g1 = x[1] - ind[2], g2 = x[2] - ind[3];
    const double pl[3] = {R[0] * g0 + R[3] * g1 + R[6] * g2, R[1] * g0 + R[4] * g1 + R[7] * g2, R[2] * g0 + R[5] * g1 + R[8] * g2};  // R^T (x - c)
    // need_distance = false (energy evaluations): anything at or beyond d_hat is as good as infinitely far
    const double reach = m.dhat + ind[4];
    const MeshDist md = mesh_distance(m.im_nt, m.im_tri, m.im_bs, m.im_cl, pl[0], pl[1], pl[2], need_distance ? 1e300 : reach * reach * (1.0 + 1e-12));
    const double nl[3] = {md.n0, md.n1, md.n2};
    c.d = md.d - ind[4];
    c.n[0] = R[0] * nl[0] + R[1] * nl[1] + R[2] * nl[2];
    c.n[1] = R[3] * nl[0] + R[4] * nl[1] + R[5] * nl[2];
    c.n[2] = R[6] * nl[0] + R[7] * nl[1] + R[8] * nl[2];
  } else {
    return c;
  }
  if (c.d <= 0.0) { c.penetrating = true; c.e = INFINITY; return c; }
  if (c.d >= m.dhat) return c;
  c.active = true;
  const double sg = c.d / m.dhat, ln = log(sg), q = sg - 1.0, kw = m.kappa * w;
  c.e = -kw * q * q * ln;
  c.b1 = kw * (-2.0 * q * ln - q * q / sg) / m.dhat;
  c.b2 = kw * (-2.0 * ln - 4.0 * q / sg + q * q / (sg * sg)) / (m.dhat * m.dhat);
  return c;
}
// ---- lagged Coulomb friction of one surface vertex (IPC, Li et al. 2020 eq. 18-20; US:103-124 enable_friction / friction ratio /
// eps_velocity).  Normal force lam = -dB/dd and contact normal n are LAGGED (frozen), which makes the potential a smooth function of
// x.  WHERE the lag is taken: the state the step starts from (the default since round 4, `lag_at_start` in fem_newton_lds_kernel) - IPC's
// lag "from the previous time step".  After the indenter has moved, that state sits deep in the 10 GPa barrier, where -dB/dd is orders of
// magnitude above the elastic forces of the soft pad (Newton directions of metres, PCG at its cap: why rounds 3-4 ran the loop in TWO
// PHASES - normal contact alone until converged, then the lag from that state and a friction phase); but the lag takes the SMALLER of
// -dB/dd and the contact REACTION (g_other . n) / dt^2, and at the start state - the previous step's equilibrium - that reaction is the
// previous step's normal force.  With the cap the start-of-step lag is well behaved, and the step saves the iteration the second phase
// cost (a pressing step is one Newton iteration instead of two).  TACEX_FEM_FRIC_LAG=0 keeps the two-phase loop for the A/B.
// u = (I - n n^T)(x - x_n - disp) is the tangential sliding relative to the indenter (x_n = positions the step
// started from, disp = the indenter's own displacement since the previous step).  Potential mu lam f0(|u|), f0(y) = -y^3 / (3 eps^2) + y^2 / eps + eps / 3 below the stick
// tolerance eps, y beyond; gradient mu lam (f1 / y) u; Hessian mu lam [(f1 / y)(T - t t^T) + f1' t t^T] (both coefficients >= 0).
struct FricVertex {  // what a vertex keeps in LDS for the step: lam, n (4 doubles)
  double lam, n[3];
};
struct FricEval {
  double e;       // mu lam f0(y)              (NOT times dt^2)
  double g[3];    // gradient
  double h[6];    // Hessian, symmetric: xx xy xz yy yz zz
};
__device__ __forceinline__ FricEval friction_eval(double mu, double eps, const double* fv /* lam, n */, const double x[3], const double xn[3],
                                                  const double disp[3], bool with_hessian) {
  FricEval f;
  f.e = 0.0; f.g[0] = f.g[1] = f.g[2] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) f.h[k] = 0.0;
  const double lam = fv[0];
  if (!(lam > 0.0)) return f;
  const double n0 = fv[1], n1 = fv[2], n2 = fv[3];
  const double r0 = x[0] - xn[0] - disp[0], r1 = x[1] - xn[1] - disp[1], r2 = x[2] - xn[2] - disp[2];
  const double rn = r0 * n0 + r1 * n1 + r2 * n2;
  const double u0 = r0 - rn * n0, u1 = r1 - rn * n1, u2 = r2 - rn * n2;
  const double y = sqrt(u0 * u0 + u1 * u1 + u2 * u2);
  const bool stick = y < eps;
  const double a = stick ? 2.0 / eps - y / (eps * eps) : 1.0 / y;   // f1 / y
  const double c = mu * lam;
  f.e = c * (stick ? -y * y * y / (3.0 * eps * eps) + y * y / eps + eps / 3.0 : y);
  f.g[0] = c * a * u0; f.g[1] = c * a * u1; f.g[2] = c * a * u2;
  if (with_hessian) {
    const double bq = stick ? 2.0 / eps - 2.0 * y / (eps * eps) : 0.0;  // f1'
    const double iy = y > 0.0 ? 1.0 / y : 0.0;
    const double t0 = u0 * iy, t1 = u1 * iy, t2 = u2 * iy;
    const double ca = c * a, cb = c * (bq - a);  // a (T - t t^T) + bq t t^T = a T + (bq - a) t t^T
    f.h[0] = ca * (1.0 - n0 * n0) + cb * t0 * t0; f.h[1] = ca * (-n0 * n1) + cb * t0 * t1; f.h[2] = ca * (-n0 * n2) + cb * t0 * t2;
    f.h[3] = ca * (1.0 - n1 * n1) + cb * t1 * t1; f.h[4] = ca * (-n1 * n2) + cb * t1 * t2; f.h[5] = ca * (1.0 - n2 * n2) + cb * t2 * t2;
  }
  return f;
}
constexpr double kCcdSlack = 0.9;  // fraction of the conservative (1-Lipschitz) step bound d / |dx| a Newton step may use

// ---- small dense helpers (row-major 3x3 in double[9]) ---------------------------------------------------
__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

struct TetState {
  double F[9], C[9];
  double a, b, c;   // coefficients above
  double Ic, J;
};

// the same through the AoS record (see FemDev::tet_rec); also returns the volume
__device__ __forceinline__ void load_tet_rec(const FemDev& m, int t, int v[4], double Di[9], double& vol) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d* q = reinterpret_cast<const v2d*>(m.tet_rec + (size_t)t * 12);
  const v2d q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5];
  v[0] = __double2loint(q0.x); v[1] = __double2hiint(q0.x); v[2] = __double2loint(q0.y); v[3] = __double2hiint(q0.y);
  Di[0] = q1.x; Di[1] = q1.y; Di[2] = q2.x; Di[3] = q2.y; Di[4] = q3.x; Di[5] = q3.y; Di[6] = q4.x; Di[7] = q4.y; Di[8] = q5.x;
  vol = q5.y;
}

// table read at (uniform base) + (32-bit byte offset): selects the scalar-base form of the load (global_load v, v_off, s[base:base+1]), so
// a loop keeps ONE 32-bit offset alive instead of a 64-bit per-lane address per table - the Newton kernel carried ~25 such addresses
// across its PCG loop, spilled them, and read them back from scratch (which misses the L2: 512 envs x 300 KB) one dependent wait at a time
template <typename T>
__device__ __forceinline__ T ldg_off(const void* base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}

// a value the optimiser must treat as unknown: address arithmetic built on it is recomputed where it is used (a few integer
// operations) instead of being hoisted out of the enclosing loops and kept live - or spilled - across them
__device__ __forceinline__ unsigned opaque_u32(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

// The thread index rebuilt from nothing but the wave's index (a scalar register) and the lane counter: inside the Newton kernel the
// register allocator spilled threadIdx.x itself - and the LDS addresses derived from it - and re-read them from scratch fourteen
// times per PCG iteration.  volatile: every call site gets its own two-instruction copy, nothing is carried between phases.
__device__ __forceinline__ int fresh_tid(int wave_index) {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return wave_index * 64 + l;
}

constexpr unsigned kTetBlkBytes = 9 * 512 + 512 + 4 * 256;  // 6144
// one tet through the wave-blocked table (see FemDev::tet_blk): coalesced like the SoA arrays, one offset register
__device__ __forceinline__ void load_tet_blk(const FemDev& m, int t, int v[4], double Di[9], double& vol) {
  const unsigned ln = (unsigned)t & 63u;
  const unsigned ob = ((unsigned)t >> 6) * kTetBlkBytes;
  const char* base = reinterpret_cast<const char*>(m.tet_blk);
  const unsigned o8 = ob + ln * 8u, o4 = ob + 5120u + ln * 4u;
#pragma unroll
  for (int k = 0; k < 9; ++k) Di[k] = *reinterpret_cast<const double*>(base + (o8 + (unsigned)k * 512u));
  vol = *reinterpret_cast<const double*>(base + (o8 + 4608u));
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const int*>(base + (o4 + (unsigned)k * 256u));
}

__device__ __forceinline__ void load_tet(const FemDev& m, int t, int v[4], double Di[9]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = m.tets[k * m.T + t];
#pragma unroll
  for (int k = 0; k < 9; ++k) Di[k] = m.dminv[k * m.T + t];
}

// F = Ds * DmInv with Ds columns (x1-x0, x2-x0, x3-x0); x points at one env's (V,3) array
__device__ __forceinline__ void deformation_gradient(const double* x, const int v[4], const double Di[9], double F[9]) {
  double Ds[9];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) Ds[i * 3 + k] = x[v[k + 1] * 3 + i] - x[v[0] * 3 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int mm = 0; mm < 3; ++mm)
      F[i * 3 + mm] = Ds[i * 3 + 0] * Di[0 * 3 + mm] + Ds[i * 3 + 1] * Di[1 * 3 + mm] + Ds[i * 3 + 2] * Di[2 * 3 + mm];
}

__device__ __forceinline__ void tet_state(const FemDev& m, const double F[9], TetState& s) {
  double f0[3] = {F[0], F[3], F[6]}, f1[3] = {F[1], F[4], F[7]}, f2[3] = {F[2], F[5], F[8]};
  double c0[3], c1[3], c2[3];
  cross3(f1, f2, c0);
  cross3(f2, f0, c1);
  cross3(f0, f1, c2);
#pragma unroll
  for (int i = 0; i < 3; ++i) { s.C[i * 3 + 0] = c0[i]; s.C[i * 3 + 1] = c1[i]; s.C[i * 3 + 2] = c2[i]; }
  double Ic = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) { s.F[k] = F[k]; Ic += F[k] * F[k]; }
  s.Ic = Ic;
  s.J = f0[0] * c0[0] + f0[1] * c0[1] + f0[2] * c0[2];
  s.a = m.mu * (1.0 - 1.0 / (Ic + 1.0));
  s.b = 2.0 * m.mu / ((Ic + 1.0) * (Ic + 1.0));
  s.c = m.lam * (s.J - m.alpha);
}

__device__ __forceinline__ double psi_of(const FemDev& m, const TetState& s) {
  const double dj = s.J - m.alpha;
  return 0.5 * m.mu * (s.Ic - 3.0) + 0.5 * m.lam * dj * dj - 0.5 * m.mu * log(s.Ic + 1.0) - m.psi_rest;
}

// dP = (9x9 Hessian of Psi) applied to dF
__device__ __forceinline__ void apply_dP(const FemDev& m, const TetState& s, const double dF[9], double dP[9]) {
  double FdF = 0.0, CdF = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) { FdF += s.F[k] * dF[k]; CdF += s.C[k] * dF[k]; }
  const double* F = s.F;
  double f0[3] = {F[0], F[3], F[6]}, f1[3] = {F[1], F[4], F[7]}, f2[3] = {F[2], F[5], F[8]};
  double d0[3] = {dF[0], dF[3], dF[6]}, d1[3] = {dF[1], dF[4], dF[7]}, d2[3] = {dF[2], dF[5], dF[8]};
  double t1[3], t2[3], e0[3], e1[3], e2[3];
  cross3(d1, f2, t1); cross3(f1, d2, t2);
#pragma unroll
  for (int i = 0; i < 3; ++i) e0[i] = t1[i] + t2[i];
  cross3(d2, f0, t1); cross3(f2, d0, t2);
#pragma unroll
  for (int i = 0; i < 3; ++i) e1[i] = t1[i] + t2[i];
  cross3(d0, f1, t1); cross3(f0, d1, t2);
#pragma unroll
  for (int i = 0; i < 3; ++i) e2[i] = t1[i] + t2[i];
  const double bb = s.b * FdF, ll = m.lam * CdF;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dP[i * 3 + 0] = s.a * dF[i * 3 + 0] + bb * F[i * 3 + 0] + ll * s.C[i * 3 + 0] + s.c * e0[i];
    dP[i * 3 + 1] = s.a * dF[i * 3 + 1] + bb * F[i * 3 + 1] + ll * s.C[i * 3 + 1] + s.c * e1[i];
    dP[i * 3 + 2] = s.a * dF[i * 3 + 2] + bb * F[i * 3 + 2] + ll * s.C[i * 3 + 2] + s.c * e2[i];
  }
}

// rows r_v (v = 0..3) with dF[k][m] / dx[v][k] = r_v[m]:  r_{1..3} = rows of DmInv, r_0 = -(r_1 + r_2 + r_3)
__device__ __forceinline__ void shape_rows(const double Di[9], double r[12]) {
#pragma unroll
  for (int mm = 0; mm < 3; ++mm) {
    r[3 + mm] = Di[0 * 3 + mm]; r[6 + mm] = Di[1 * 3 + mm]; r[9 + mm] = Di[2 * 3 + mm];
    r[mm] = -(Di[0 * 3 + mm] + Di[1 * 3 + mm] + Di[2 * 3 + mm]);
  }
}

// element gradient (12) = scale * P : dF/dx
__device__ __forceinline__ void element_gradient(const TetState& s, const double r[12], double scale, double g[12]) {
  double P[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) P[k] = s.a * s.F[k] + s.c * s.C[k];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      g[v * 3 + i] = scale * (P[i * 3 + 0] * r[v * 3 + 0] + P[i * 3 + 1] * r[v * 3 + 1] + P[i * 3 + 2] * r[v * 3 + 2]);
}

// cyclic Jacobi eigen-decomposition of a symmetric 9x9 (PSD-projection path only; arrays live in scratch)
__device__ void jacobi_psd9(double* A) {
  double V[81];
  for (int i = 0; i < 81; ++i) V[i] = (i / 9 == i % 9) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int p = 0; p < 9; ++p) {
      dia += A[p * 9 + p] * A[p * 9 + p];
      for (int q = p + 1; q < 9; ++q) off += A[p * 9 + q] * A[p * 9 + q];
    }
    if (off <= 1e-30 * (dia + 1e-300)) break;
    for (int p = 0; p < 8; ++p)
      for (int q = p + 1; q < 9; ++q) {
        const double apq = A[p * 9 + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[q * 9 + q] - A[p * 9 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 9; ++k) {
          const double akp = A[k * 9 + p], akq = A[k * 9 + q];
          A[k * 9 + p] = cs * akp - sn * akq;
          A[k * 9 + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 9; ++k) {
          const double apk = A[p * 9 + k], aqk = A[q * 9 + k];
          A[p * 9 + k] = cs * apk - sn * aqk;
          A[q * 9 + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < 9; ++k) {
          const double vkp = V[k * 9 + p], vkq = V[k * 9 + q];
          V[k * 9 + p] = cs * vkp - sn * vkq;
          V[k * 9 + q] = sn * vkp + cs * vkq;
        }
      }
  }
  double w[9];
  for (int i = 0; i < 9; ++i) w[i] = A[i * 9 + i] > 0.0 ? A[i * 9 + i] : 0.0;
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) {
      double sacc = 0.0;
      for (int k = 0; k < 9; ++k) sacc += V[i * 9 + k] * w[k] * V[j * 9 + k];
      A[i * 9 + j] = sacc;
    }
}

// ---- K17a: element terms, one tet per lane, SoA outputs ---------------------------------------------------
template <bool PROJECT_PSD>
__global__ __launch_bounds__(256) void fem_element_terms_kernel(FemDev m, const double* __restrict__ x,
                                                                double* __restrict__ energy, double* __restrict__ grad,
                                                                double* __restrict__ hess) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= m.T) return;
  int v[4];
  double Di[9], F[9], r[12];
  load_tet(m, t, v, Di);
  deformation_gradient(x + (size_t)b * m.V * 3, v, Di, F);
  TetState s;
  tet_state(m, F, s);
  shape_rows(Di, r);
  const double vol = m.vol[t];
  const size_t T = m.T;
  if (energy) energy[(size_t)b * T + t] = vol * psi_of(m, s);
  if (grad) {
    double g[12];
    element_gradient(s, r, vol, g);
#pragma unroll
    for (int k = 0; k < 12; ++k) grad[((size_t)b * 12 + k) * T + t] = g[k];
  }
  if (!hess) return;
  if constexpr (!PROJECT_PSD) {
    // column j = (vertex u, component k): dF = e_k (x) r_u ; H[:, j] = vol * (dP : dF_i)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double dF[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dP[9];
        dF[k * 3 + 0] = r[u * 3 + 0]; dF[k * 3 + 1] = r[u * 3 + 1]; dF[k * 3 + 2] = r[u * 3 + 2];
        apply_dP(m, s, dF, dP);
        const int j = u * 3 + k;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double h = vol * (dP[i * 3 + 0] * r[w * 3 + 0] + dP[i * 3 + 1] * r[w * 3 + 1] + dP[i * 3 + 2] * r[w * 3 + 2]);
            hess[((size_t)b * 144 + (w * 3 + i) * 12 + j) * T + t] = h;
          }
      }
    return;
  } else {
  // PSD projection of the 9x9 F-space Hessian (row-major vec(F) index q = i*3 + m), then H12 = vol G^T H9+ G
  double H9[81];
  for (int q = 0; q < 9; ++q) {
    double dF[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dP[9];
    dF[q] = 1.0;
    apply_dP(m, s, dF, dP);
    for (int p = 0; p < 9; ++p) H9[p * 9 + q] = dP[p];
  }
  for (int p = 0; p < 9; ++p)
    for (int q = p + 1; q < 9; ++q) { const double a = 0.5 * (H9[p * 9 + q] + H9[q * 9 + p]); H9[p * 9 + q] = a; H9[q * 9 + p] = a; }
  jacobi_psd9(H9);
  for (int u = 0; u < 4; ++u)
    for (int k = 0; k < 3; ++k) {
      double dP[9];  // H9 * vec(dF_j), dF_j = e_k (x) r_u
      for (int p = 0; p < 9; ++p)
        dP[p] = H9[p * 9 + k * 3 + 0] * r[u * 3 + 0] + H9[p * 9 + k * 3 + 1] * r[u * 3 + 1] + H9[p * 9 + k * 3 + 2] * r[u * 3 + 2];
      const int j = u * 3 + k;
      for (int w = 0; w < 4; ++w)
        for (int i = 0; i < 3; ++i) {
          const double h = vol * (dP[i * 3 + 0] * r[w * 3 + 0] + dP[i * 3 + 1] * r[w * 3 + 1] + dP[i * 3 + 2] * r[w * 3 + 2]);
          hess[((size_t)b * 144 + (w * 3 + i) * 12 + j) * T + t] = h;
        }
    }
  }
}

// ---- block-wide sum (wave shuffle + LDS), result broadcast to all threads ------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sh /* >= 17 doubles */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();  // protect sh from the previous use
  if ((threadIdx.x & 63) == 0) sh[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += sh[w];
    sh[16] = s;
  }
  __syncthreads();
  return sh[16];
}

// block-wide maximum, same scheme
__device__ __forceinline__ double block_sum_max(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = sh[0];
    for (int w = 1; w < nw; ++w) s = fmax(s, sh[w]);
    sh[16] = s;
  }
  __syncthreads();
  return sh[16];
}
constexpr int kLsRescueStream = 32;  // rescue halvings of the line search next to a barrier (kLsRescue of the CU-resident kernel)

__device__ double env_energy(const FemDev& m, const double* x, const double* xt, const uint8_t* cons, const double* aim,
                             double* sh, const double* ind = nullptr, const double* fl = nullptr, const double* xn = nullptr,
                             const double* disp = nullptr) {
  double e = 0.0;
  for (int t = threadIdx.x; t < m.T; t += blockDim.x) {
    int v[4];
    double Di[9], F[9];
    load_tet(m, t, v, Di);
    deformation_gradient(x, v, Di, F);
    TetState s;
    tet_state(m, F, s);
    e += m.dt * m.dt * m.vol[t] * psi_of(m, s);
  }
  for (int v = threadIdx.x; v < m.V; v += blockDim.x) {
    const double mv = m.mass[v];
    double q = 0.0, qc = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double d = x[v * 3 + i] - xt[v * 3 + i];
      q += d * d;
      if (cons && cons[v]) { const double c = x[v * 3 + i] - aim[v * 3 + i]; qc += c * c; }
    }
    e += 0.5 * mv * q + 0.5 * m.strength * mv * qc;
    if (ind && m.area) e += m.dt * m.dt * contact_eval(m, ind, m.area[v], x + v * 3).e;
    if (fl) e += m.dt * m.dt * friction_eval(m.fric_mu, m.fric_eps, fl + (size_t)v * 4, x + v * 3, xn + v * 3, disp, false).e;
  }
  return block_sum(e, sh);
}

// per-tet gradients (scaled by dt^2) into ge (12,T) of this env
__device__ void env_tet_gradients(const FemDev& m, const double* x, double* ge) {
  for (int t = threadIdx.x; t < m.T; t += blockDim.x) {
    int v[4];
    double Di[9], F[9], r[12], g[12];
    load_tet(m, t, v, Di);
    deformation_gradient(x, v, Di, F);
    TetState s;
    tet_state(m, F, s);
    shape_rows(Di, r);
    element_gradient(s, r, m.dt * m.dt * m.vol[t], g);
#pragma unroll
    for (int k = 0; k < 12; ++k) ge[(size_t)k * m.T + t] = g[k];
  }
}

// atomics-free nodal assembly: vertex v sums its incident tets' local rows
__device__ __forceinline__ void gather_vertex(const FemDev& m, const double* ge, int v, double out[3]) {
  out[0] = out[1] = out[2] = 0.0;
  for (int e = m.vt_off[v]; e < m.vt_off[v + 1]; ++e) {
    const int code = m.vt_idx[e];
    const int t = code >> 2, l = code & 3;
    out[0] += ge[(size_t)(l * 3 + 0) * m.T + t];
    out[1] += ge[(size_t)(l * 3 + 1) * m.T + t];
    out[2] += ge[(size_t)(l * 3 + 2) * m.T + t];
  }
}

__global__ __launch_bounds__(512) void fem_energy_kernel(FemDev m, const double* x, const double* xt,
                                                         const uint8_t* cons, const double* aim, double* E) {
  __shared__ double sh[17];
  const int b = blockIdx.x;
  const size_t o = (size_t)b * m.V * 3;
  const double e = env_energy(m, x + o, xt + o, cons ? cons + (size_t)b * m.V : nullptr, aim ? aim + o : nullptr, sh,
                              m.indenters ? m.indenters + (size_t)b * 8 : nullptr);
  if (threadIdx.x == 0) E[b] = e;
}

__global__ __launch_bounds__(512) void fem_gradient_kernel(FemDev m, const double* x, const double* xt,
                                                           const uint8_t* cons, const double* aim, double* g,
                                                           double* ws_ge /* (B,12,T) */) {
  const int b = blockIdx.x;
  const size_t o = (size_t)b * m.V * 3;
  double* ge = ws_ge + (size_t)b * 12 * m.T;
  env_tet_gradients(m, x + o, ge);
  __syncthreads();
  for (int v = threadIdx.x; v < m.V; v += blockDim.x) {
    double a[3];
    gather_vertex(m, ge, v, a);
    const double mv = m.mass[v];
    const bool c = cons && cons[(size_t)b * m.V + v];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double gi = a[i] + mv * (x[o + v * 3 + i] - xt[o + v * 3 + i]);
      if (c) gi += m.strength * mv * (x[o + v * 3 + i] - aim[o + v * 3 + i]);
      g[o + v * 3 + i] = gi;
    }
    if (m.indenters && m.area) {
      const ContactEval ce = contact_eval(m, m.indenters + (size_t)b * 8, m.area[v], x + o + v * 3);
      if (ce.active)
#pragma unroll
        for (int i = 0; i < 3; ++i) g[o + v * 3 + i] += m.dt * m.dt * ce.b1 * ce.n[i];
    }
  }
}

// ---- K17b: one projected-Newton iteration per env, everything inside one workgroup -----------------------------
// workspace per env (doubles): ge 12T | tet cache 12T (F 9, a, b, c) | hv 12T | g,r,z,p,d,Hp,xc 7*3V | Dinv 9V | contact 5V | friction lag 4V | friction blocks 6V; behind the B env
// blocks: x_prev (B,V,3) and the per-env max |d| of tacex_fem_step
__host__ __device__ inline size_t newton_ws_doubles(int V, int T) { return (size_t)36 * T + (size_t)45 * V; }  // (+ 10 V: friction lag | Hessian blocks)

__device__ __forceinline__ bool inv3_spd(const double A[9], double Ai[9]) {
  // Cholesky test + inverse via adjugate
  if (!(A[0] > 0.0)) return false;
  const double l10 = A[3] / sqrt(A[0]), l20 = A[6] / sqrt(A[0]);
  const double d1 = A[4] - l10 * l10;
  if (!(d1 > 0.0)) return false;
  const double l21 = (A[7] - l20 * l10) / sqrt(d1);
  const double d2 = A[8] - l20 * l20 - l21 * l21;
  if (!(d2 > 0.0)) return false;
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  const double id = 1.0 / det;
  Ai[0] = c00 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  Ai[3] = c01 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  Ai[6] = c02 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  return true;
}

// Additive coarse correction of the two-level preconditioner for kernels whose vectors live in memory (fem_newton_kernel, fem_ball_newton_kernel):
// z += P A_c^-1 P^T r over the (V,3) rows of one env (tacex_fem_set_coarse_space: trilinear hats of a coarse grid, A_c the rest-state
// operator's Galerkin product).  Every thread of the workgroup calls it; rc / yc: 3 * kFemMaxCoarse doubles of LDS each.  Returns this
// thread's share of r . (P A_c^-1 P^T r) (add it to the partial sum of r . z before the block reduction).  Fixed summation order.
__device__ __forceinline__ double coarse_correct(const FemDev& m, const double* r, double* z, double* rc, double* yc) {
  const int nc3 = 3 * m.nc, NT = (int)blockDim.x, tid = (int)threadIdx.x;
  int G = 1;
  while (2 * G <= NT / m.nc && 2 * G <= 64) G *= 2;
  const int node = tid / G, j = tid - node * G;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  if (node < m.nc) {
    const int e1 = m.cn_off[node + 1];
    for (int e = m.cn_off[node] + j; e < e1; e += G) {
      const int v0 = m.cn_vtx[e];
      const double w0 = m.cn_w[e];
      a0 += w0 * r[v0 * 3]; a1 += w0 * r[v0 * 3 + 1]; a2 += w0 * r[v0 * 3 + 2];
    }
  }
  for (int o2 = G >> 1; o2 > 0; o2 >>= 1) { a0 += __shfl_xor(a0, o2, 64); a1 += __shfl_xor(a1, o2, 64); a2 += __shfl_xor(a2, o2, 64); }
  if (node < m.nc && j == 0) { rc[node * 3] = a0; rc[node * 3 + 1] = a1; rc[node * 3 + 2] = a2; }
  __syncthreads();
  double part = 0.0;
  if (tid < nc3) {
    double sv = 0.0;
    for (int k = 0; k < nc3; ++k) sv += m.ac_inv[(size_t)tid * nc3 + k] * rc[k];
    yc[tid] = sv;
    part = rc[tid] * sv;
  }
  __syncthreads();
  for (int v = tid; v < m.V; v += NT) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int nd = m.cv_node[v * 8 + k];
      const double w = m.cv_w[v * 8 + k];
      z[v * 3] += w * yc[nd * 3]; z[v * 3 + 1] += w * yc[nd * 3 + 1]; z[v * 3 + 2] += w * yc[nd * 3 + 2];
    }
  }
  return part;
}

// flags of step_info[., 2]
constexpr int kFemFlagPenetration = 1;  // a contact vertex was at or beyond the indenter surface when the iteration started
constexpr int kFemFlagLsFailed = 2;     // a line search found no decrease even after the rescue halvings
constexpr int kFemFlagCoarseOff = 4;    // informational: the coarse correction was switched off for the rest of the step (see kCoarseTrust)
constexpr int kFemFlagPsdSafe = 8;      // informational: the PCG met negative curvature and the env solved iterations of the step in PSD-safe mode
__global__ __launch_bounds__(512) void fem_newton_kernel(FemDev m, double* xg, const double* xtg, const uint8_t* consg,
                                                         const double* aimg, double* stats, double* wsg,
                                                         int pcg_max_iter, double pcg_tol_rate, int ls_max_iter, double* dxg, double dx_tol,
                                                         double* step_info, int accumulate, const double* xprevg, const double* dispg,
                                                         int fric_ipc, int lds_sweep) {
  // step_info (nullable): the row of this env [Newton iterations, max |d|, flags, PCG iterations] - SET by the first launch of a time
  // step (accumulate = 0), added to / OR-ed by the later ones: tacex_fem_step runs this kernel once per Newton iteration, and
  // UipcSim.check_step() must see a penetrating vertex or a dead line search of ANY of them (the row used to be zeroed)
  __shared__ double sh[17];
  __shared__ double crc[3 * kFemMaxCoarse], cyc[3 * kFemMaxCoarse];
  // lds_sweep (round 6; atomic mode only, 9 V doubles of dynamic LDS): x, the PCG direction p and per-vertex H.p accumulators live in LDS - a
  // tet recomputes its state from x, gathers p from LDS and ADDS its rows with ds_add_f64, instead of reading a cached state (96 B) and
  // writing rows (96 B) through HBM and a CSR gather afterwards: what made this kernel 4x slower per iteration than the CU-resident one.
  // The deterministic switch keeps the fixed-order gather (lds_sweep = 0).
  extern __shared__ __attribute__((aligned(16))) double nws_lds[];
  const int b = blockIdx.x;
  if (dxg && dxg[b] <= dx_tol) {  // converged in an earlier launch of this time step (same protocol as the CU-resident kernel)
    if (threadIdx.x == 0) { stats[(size_t)b * 4 + 2] = 0.0; stats[(size_t)b * 4 + 3] = 0.0; }
    return;
  }
  const int V = m.V, T = m.T;
  const size_t o = (size_t)b * V * 3;
  double* x = xg + o;
  const double* xt = xtg + o;
  const uint8_t* cons = consg ? consg + (size_t)b * V : nullptr;
  const double* aim = aimg ? aimg + o : nullptr;
  double* ws = wsg + (size_t)b * newton_ws_doubles(V, T);
  double* ge = ws;                       // (12,T) tet gradients, later diag-block scratch
  double* tc = ws + (size_t)12 * T;      // (12,T) F(9), a, b, c
  double* hv = ws + (size_t)24 * T;      // (12,T) per-tet H*p contributions
  double* vg = ws + (size_t)36 * T;      // g
  // lds_sweep = 2: the PCG's r, z, d and H.p live in LDS as well (21 V doubles: 120 KB at 715 vertices) - the direction is computed and
  // consumed (step bound, line search) inside this launch, so nothing of the loop has to pass through memory (the workspace slots stay
  // where they are: the layout is one)
  double* vr = lds_sweep == 2 ? nws_lds + 9 * V : vg + (size_t)3 * V;
  double* vz = lds_sweep == 2 ? nws_lds + 12 * V : vg + (size_t)6 * V;
  double* vp = lds_sweep ? nws_lds + 3 * V : vg + (size_t)9 * V;
  double* xs_l = nws_lds;                   // (V,3) x
  double* acc_l = nws_lds + 6 * V;          // (V,3) accumulators
  double* vd = lds_sweep == 2 ? nws_lds + 15 * V : vg + (size_t)12 * V;
  double* vHp = lds_sweep == 2 ? nws_lds + 18 * V : vg + (size_t)15 * V;
  double* xc = vg + (size_t)18 * V;      // line-search candidate
  double* Dinv = xc + (size_t)3 * V;     // (V,9)
  double* cdat = Dinv + (size_t)9 * V;   // (V,5) barrier of the vertex at x: dt^2 b'' | n (3) | gap d
  double* flag_ = cdat + (size_t)5 * V;  // (V,4) friction lag: normal force | normal - taken in the FIRST launch of a time step, kept for its others
  double* fhs = flag_ + (size_t)4 * V;   // (V,6) friction Hessian block of the vertex at x (dt^2-scaled, rounded to float like the CU-resident kernel's)
  const double dt2 = m.dt * m.dt;
  // IPC barrier against the env's indenter (the same terms as in the CU-resident kernel: gradient b' n, PSD curvature b'' n n^T in
  // H.p and the block-Jacobi blocks, conservative step bound before the line search) and, since round 5, Coulomb friction with the lag
  // taken at the start of the step (capped by the contact reaction, or IPC's previous-configuration lag: tacex_fem_set_friction_lag).
  // Vertex chains, the coarse correction, the contact-following start and the edge snap exist in the CU-resident kernel only - this
  // is the path of meshes with more vertices than its workgroup has threads.
  const double* ind = (m.indenters && m.area) ? m.indenters + (size_t)b * 8 : nullptr;
  const bool fric = ind && m.fric_mu > 0.0 && xprevg != nullptr && dispg != nullptr;
  const double* xn = fric ? xprevg + o : nullptr;
  double disp3[3] = {0, 0, 0};
  if (fric) { disp3[0] = dispg[b * 3]; disp3[1] = dispg[b * 3 + 1]; disp3[2] = dispg[b * 3 + 2]; }
  const double* fl = fric ? flag_ : nullptr;

  // ---- element pass: cache F and coefficients, tet gradients, diagonal 3x3 blocks (into hv as (4*9? no: 12 rows)) ----
  // the four 3x3 diagonal blocks of the element Hessian need 36 doubles per tet: use ge+tc? they are needed later,
  // so diagonal blocks are accumulated vertex-side from recomputed columns below (second loop) instead.
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    int v[4];
    double Di[9], F[9], r[12], g[12];
    load_tet(m, t, v, Di);
    deformation_gradient(x, v, Di, F);
    TetState s;
    tet_state(m, F, s);
    shape_rows(Di, r);
    element_gradient(s, r, dt2 * m.vol[t], g);
#pragma unroll
    for (int k = 0; k < 12; ++k) ge[(size_t)k * T + t] = g[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) tc[(size_t)k * T + t] = F[k];
    tc[(size_t)9 * T + t] = s.a; tc[(size_t)10 * T + t] = s.b; tc[(size_t)11 * T + t] = s.c;
  }
  __syncthreads();
  // ---- nodal gradient + block-Jacobi preconditioner (vertex gather; diagonal blocks recomputed per incidence) ----
  int pen = 0;  // a contact vertex of this thread sits at or beyond its indenter's surface (kFemFlagPenetration)
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    double a3[3];
    gather_vertex(m, ge, v, a3);
    const double mv = m.mass[v];
    const bool c = cons && cons[v];
    const double md = mv * (1.0 + (c ? m.strength : 0.0));
    double D[9] = {md, 0, 0, 0, md, 0, 0, 0, md};
    double cg[3] = {0, 0, 0};
    {
      const double xv[3] = {x[v * 3], x[v * 3 + 1], x[v * 3 + 2]};
      const ContactEval ce = contact_eval(m, ind, ind ? m.area[v] : 0.0, xv);
      if (ce.penetrating) pen = 1;
      const double cb2 = ce.active ? dt2 * ce.b2 : 0.0;
      cdat[(size_t)v * 5] = cb2;
      cdat[(size_t)v * 5 + 1] = ce.n[0]; cdat[(size_t)v * 5 + 2] = ce.n[1]; cdat[(size_t)v * 5 + 3] = ce.n[2];
      cdat[(size_t)v * 5 + 4] = (ind && m.area[v] > 0.0 && !ce.penetrating) ? ce.d : 1e300;
      if (ce.active) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          cg[i] = dt2 * ce.b1 * ce.n[i];
#pragma unroll
          for (int k = 0; k < 3; ++k) D[i * 3 + k] += cb2 * ce.n[i] * ce.n[k];
        }
      }
    }
    for (int e = m.vt_off[v]; e < m.vt_off[v + 1]; ++e) {
      const int code = m.vt_idx[e];
      const int t = code >> 2, l = code & 3;
      double Di[9], r[12];
#pragma unroll
      for (int k = 0; k < 9; ++k) Di[k] = m.dminv[(size_t)k * T + t];
      shape_rows(Di, r);
      TetState s;
#pragma unroll
      for (int k = 0; k < 9; ++k) s.F[k] = tc[(size_t)k * T + t];
      {  // cofactor from F
        double f0[3] = {s.F[0], s.F[3], s.F[6]}, f1[3] = {s.F[1], s.F[4], s.F[7]}, f2[3] = {s.F[2], s.F[5], s.F[8]};
        double c0[3], c1[3], c2[3];
        cross3(f1, f2, c0); cross3(f2, f0, c1); cross3(f0, f1, c2);
#pragma unroll
        for (int i = 0; i < 3; ++i) { s.C[i * 3 + 0] = c0[i]; s.C[i * 3 + 1] = c1[i]; s.C[i * 3 + 2] = c2[i]; }
      }
      s.a = tc[(size_t)9 * T + t]; s.b = tc[(size_t)10 * T + t]; s.c = tc[(size_t)11 * T + t];
      const double sc = dt2 * m.vol[t];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double dF[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dP[9];
        dF[k * 3 + 0] = r[l * 3 + 0]; dF[k * 3 + 1] = r[l * 3 + 1]; dF[k * 3 + 2] = r[l * 3 + 2];
        apply_dP(m, s, dF, dP);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          D[i * 3 + k] += sc * (dP[i * 3 + 0] * r[l * 3 + 0] + dP[i * 3 + 1] * r[l * 3 + 1] + dP[i * 3 + 2] * r[l * 3 + 2]);
      }
    }
    double fg[3] = {0, 0, 0};
    if (fric) {
      const double xv[3] = {x[v * 3], x[v * 3 + 1], x[v * 3 + 2]};
      if (!accumulate) {  // first launch of the time step: the lag (FrictionModel.update of the oracle; fem_newton_lds_kernel's lag_pending)
        double lam = 0.0, ln[3] = {cdat[(size_t)v * 5 + 1], cdat[(size_t)v * 5 + 2], cdat[(size_t)v * 5 + 3]};
        if (fric_ipc) {
          double indp[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) indp[k] = ind[k];
          indp[1] -= disp3[0]; indp[2] -= disp3[1]; indp[3] -= disp3[2];
          const double xn3[3] = {xn[v * 3], xn[v * 3 + 1], xn[v * 3 + 2]};
          const ContactEval cp = contact_eval(m, indp, m.area[v], xn3);
          lam = (cp.active && !cp.penetrating) ? -cp.b1 : 0.0;
          ln[0] = cp.n[0]; ln[1] = cp.n[1]; ln[2] = cp.n[2];
        } else {
          const ContactEval ce = contact_eval(m, ind, m.area[v], xv);
          if (ce.active) {
            double go[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              go[i] = a3[i] + mv * (xv[i] - xt[v * 3 + i]);
              if (c) go[i] += m.strength * mv * (xv[i] - aim[v * 3 + i]);
            }
            const double react = (go[0] * ce.n[0] + go[1] * ce.n[1] + go[2] * ce.n[2]) / dt2;
            lam = fmin(-ce.b1, fmax(react, 0.0));
            ln[0] = ce.n[0]; ln[1] = ce.n[1]; ln[2] = ce.n[2];
          }
        }
        const bool on = lam > 0.0;
        flag_[(size_t)v * 4] = on ? lam : 0.0;
        flag_[(size_t)v * 4 + 1] = on ? ln[0] : 0.0; flag_[(size_t)v * 4 + 2] = on ? ln[1] : 0.0; flag_[(size_t)v * 4 + 3] = on ? ln[2] : 0.0;
      }
      const double xn3[3] = {xn[v * 3], xn[v * 3 + 1], xn[v * 3 + 2]};
      const FricEval fe = friction_eval(m.fric_mu, m.fric_eps, flag_ + (size_t)v * 4, xv, xn3, disp3, true);
      double h[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) { h[k] = (double)(float)(dt2 * fe.h[k]); fhs[(size_t)v * 6 + k] = h[k]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) fg[i] = dt2 * fe.g[i];
      D[0] += h[0]; D[1] += h[1]; D[2] += h[2]; D[3] += h[1]; D[4] += h[3]; D[5] += h[4]; D[6] += h[2]; D[7] += h[4]; D[8] += h[5];
    }
    double Di3[9];
    if (!inv3_spd(D, Di3)) {  // elastic block not SPD -> mass block (always SPD)
      const double im = 1.0 / md;
      Di3[0] = im; Di3[1] = 0; Di3[2] = 0; Di3[3] = 0; Di3[4] = im; Di3[5] = 0; Di3[6] = 0; Di3[7] = 0; Di3[8] = im;
    }
    // rounded to float like the blocks the CU-resident kernel keeps in LDS: the two kernels apply the SAME preconditioner
    const int up[9] = {0, 1, 2, 1, 4, 5, 2, 5, 8};
#pragma unroll
    for (int k = 0; k < 9; ++k) Dinv[(size_t)v * 9 + k] = (double)(float)Di3[up[k]];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double gi = a3[i] + mv * (x[v * 3 + i] - xt[v * 3 + i]) + cg[i] + fg[i];
      if (c) gi += m.strength * mv * (x[v * 3 + i] - aim[v * 3 + i]);
      vg[v * 3 + i] = gi;
      vr[v * 3 + i] = -gi;
      vd[v * 3 + i] = 0.0;
    }
  }
  const int any_pen = __syncthreads_or(pen);
  // z = Dinv r ; p = z ; rz
  double part = 0.0;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double z = Dinv[(size_t)v * 9 + i * 3 + 0] * vr[v * 3 + 0] + Dinv[(size_t)v * 9 + i * 3 + 1] * vr[v * 3 + 1] +
                       Dinv[(size_t)v * 9 + i * 3 + 2] * vr[v * 3 + 2];
      vz[v * 3 + i] = z;
      part += vr[v * 3 + i] * z;
    }
  }
  // two-level preconditioner (round 6): block Jacobi + the additive coarse correction of tacex_fem_set_coarse_space - what the CU-resident
  // kernel applies minus its vertex chains; meshes beyond its reach (> 768 vertices, or the deterministic switch) no longer pay block
  // Jacobi's iteration counts.  The summation order is fixed: deterministic runs stay bit-identical.
  const bool two_level = m.nc > 0 && m.cn_off && m.ac_inv;
  if (two_level) {
    __syncthreads();
    part += coarse_correct(m, vr, vz, crc, cyc);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 3 * V; k += blockDim.x) vp[k] = vz[k];
  double rz = block_sum(part, sh);
  const double rz0 = rz;
  int it = 0;
  if (lds_sweep) {
    for (int k = threadIdx.x; k < 3 * V; k += blockDim.x) { xs_l[k] = x[k]; acc_l[k] = 0.0; }
    __syncthreads();
  }
  while (it < pcg_max_iter && rz0 > 0.0 && rz > pcg_tol_rate * rz0) {  // (libuipc's test: on r.z itself, see fem_newton_lds_kernel)
    // ---- Hp = (M + s Mc + dt^2 K) p, matrix-free: per-tet dP[dF(p)] then vertex gather ----
    for (int t = threadIdx.x; lds_sweep && t < T; t += blockDim.x) {
      int v[4];
      double Di[9], F[9], dF[9], dP[9], r[12];
      load_tet(m, t, v, Di);
      deformation_gradient(xs_l, v, Di, F);
      TetState s;
      tet_state(m, F, s);
      deformation_gradient(vp, v, Di, dF);
      apply_dP(m, s, dF, dP);
      shape_rows(Di, r);
      const double sc = dt2 * m.vol[t];
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          atomicAdd(&acc_l[v[w] * 3 + i], sc * (dP[i * 3 + 0] * r[w * 3 + 0] + dP[i * 3 + 1] * r[w * 3 + 1] + dP[i * 3 + 2] * r[w * 3 + 2]));
    }
    for (int t = threadIdx.x; !lds_sweep && t < T; t += blockDim.x) {
      int v[4];
      double Di[9], dF[9], dP[9], r[12];
      load_tet(m, t, v, Di);
      deformation_gradient(vp, v, Di, dF);  // linear in p
      TetState s;
#pragma unroll
      for (int k = 0; k < 9; ++k) s.F[k] = tc[(size_t)k * T + t];
      {
        double f0[3] = {s.F[0], s.F[3], s.F[6]}, f1[3] = {s.F[1], s.F[4], s.F[7]}, f2[3] = {s.F[2], s.F[5], s.F[8]};
        double c0[3], c1[3], c2[3];
        cross3(f1, f2, c0); cross3(f2, f0, c1); cross3(f0, f1, c2);
#pragma unroll
        for (int i = 0; i < 3; ++i) { s.C[i * 3 + 0] = c0[i]; s.C[i * 3 + 1] = c1[i]; s.C[i * 3 + 2] = c2[i]; }
      }
      s.a = tc[(size_t)9 * T + t]; s.b = tc[(size_t)10 * T + t]; s.c = tc[(size_t)11 * T + t];
      apply_dP(m, s, dF, dP);
      shape_rows(Di, r);
      const double sc = dt2 * m.vol[t];
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          hv[(size_t)(w * 3 + i) * T + t] = sc * (dP[i * 3 + 0] * r[w * 3 + 0] + dP[i * 3 + 1] * r[w * 3 + 1] + dP[i * 3 + 2] * r[w * 3 + 2]);
    }
    __syncthreads();
    part = 0.0;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      double a3[3];
      if (lds_sweep) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { a3[i] = acc_l[v * 3 + i]; acc_l[v * 3 + i] = 0.0; }  // (zeroed for the next sweep by the thread that read it)
      } else {
        gather_vertex(m, hv, v, a3);
      }
      const double md = m.mass[v] * (1.0 + ((cons && cons[v]) ? m.strength : 0.0));
      const double cb2 = cdat[(size_t)v * 5];
      const double npq = cb2 * (cdat[(size_t)v * 5 + 1] * vp[v * 3] + cdat[(size_t)v * 5 + 2] * vp[v * 3 + 1] + cdat[(size_t)v * 5 + 3] * vp[v * 3 + 2]);
      double fp[3] = {0, 0, 0};
      if (fric) {
        const double* q = fhs + (size_t)v * 6;
        const double p0 = vp[v * 3], p1 = vp[v * 3 + 1], p2 = vp[v * 3 + 2];
        fp[0] = q[0] * p0 + q[1] * p1 + q[2] * p2; fp[1] = q[1] * p0 + q[3] * p1 + q[4] * p2; fp[2] = q[2] * p0 + q[4] * p1 + q[5] * p2;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double h = a3[i] + md * vp[v * 3 + i] + npq * cdat[(size_t)v * 5 + 1 + i] + fp[i];
        vHp[v * 3 + i] = h;
        part += vp[v * 3 + i] * h;
      }
    }
    const double pHp = block_sum(part, sh);
    if (!(pHp > 0.0)) {  // negative curvature: keep d (first iteration: preconditioned steepest descent)
      if (it == 0)
        for (int k = threadIdx.x; k < 3 * V; k += blockDim.x) vd[k] = vz[k];
      break;
    }
    const double al = rz / pHp;
    part = 0.0;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      double rr[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        vd[v * 3 + i] += al * vp[v * 3 + i];
        rr[i] = vr[v * 3 + i] - al * vHp[v * 3 + i];
        vr[v * 3 + i] = rr[i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double z = Dinv[(size_t)v * 9 + i * 3 + 0] * rr[0] + Dinv[(size_t)v * 9 + i * 3 + 1] * rr[1] + Dinv[(size_t)v * 9 + i * 3 + 2] * rr[2];
        vz[v * 3 + i] = z;
        part += rr[i] * z;
      }
    }
    if (two_level) {
      __syncthreads();
      part += coarse_correct(m, vr, vz, crc, cyc);
      __syncthreads();
    }
    const double rz_new = block_sum(part, sh);
    const double beta = rz_new / rz;
    for (int k = threadIdx.x; k < 3 * V; k += blockDim.x) vp[k] = vz[k] + beta * vp[k];
    rz = rz_new;
    ++it;
    __syncthreads();
  }
  __syncthreads();
  // ---- backtracking line search on the incremental potential (accept the first E(x + step d) <= E(x)) ----
  const double E0 = env_energy(m, x, xt, cons, aim, sh, ind, fl, xn, disp3);  // (lag rows: written and read by the vertex's own thread)
  double step = 1.0, E1 = E0;
  if (ind) {  // conservative step bound (1-Lipschitz distance): no surface vertex may use more than kCcdSlack of its gap
    double amax = 1.0;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      const double gap = cdat[(size_t)v * 5 + 4];
      const double nd = sqrt(vd[v * 3] * vd[v * 3] + vd[v * 3 + 1] * vd[v * 3 + 1] + vd[v * 3 + 2] * vd[v * 3 + 2]);
      if (gap < 1e299 && nd > 0.0) amax = fmin(amax, kCcdSlack * gap / nd);
    }
    step = -block_sum_max(-amax, sh);
  }
  // max |d| of the unscaled Newton direction (the convergence test; also caps the search: no vertex starts further than the body is long,
  // the direction kept on negative curvature has no length scale - same rule as the CU-resident kernel)
  double dmax = 0.0;
  for (int k = threadIdx.x; k < 3 * V; k += blockDim.x) dmax = fmax(dmax, fabs(vd[k]));
  dmax = block_sum_max(dmax, sh);
  if (dmax > m.step_cap) step = fmin(step, m.step_cap / dmax);
  bool accepted = false;
  const int ls_cap = ind ? (ls_max_iter > kLsRescueStream ? ls_max_iter : kLsRescueStream) : ls_max_iter;
  for (int ls = 0; ls <= ls_cap; ++ls) {
    for (int k = threadIdx.x; k < 3 * V; k += blockDim.x) xc[k] = x[k] + step * vd[k];
    __syncthreads();
    const double Ec = env_energy(m, xc, xt, cons, aim, sh, ind, fl, xn, disp3);
    if (Ec <= E0) { E1 = Ec; accepted = true; break; }
    step *= 0.5;
  }
  if (accepted) {
    for (int k = threadIdx.x; k < 3 * V; k += blockDim.x) x[k] = xc[k];
  } else {
    step = 0.0;
  }
  if (threadIdx.x == 0) {
    // max |d| of the unscaled direction: <= dx_tol = converged (same rule as the CU-resident kernel).  A rejected search leaves x where it
    // was - every further launch of the step would repeat this iteration: the env is taken out of the step's remaining launches (0 = done),
    // its flag says why (the CU-resident kernel and the oracle's fem_step break out of their loops the same way)
    if (dxg) dxg[b] = (step_info && !accepted && !(dmax <= dx_tol)) ? 0.0 : dmax;  // (tacex_fem_step only: tacex_fem_newton_step's callers read max |d| itself)
    stats[(size_t)b * 4 + 0] = E0; stats[(size_t)b * 4 + 1] = E1; stats[(size_t)b * 4 + 2] = step; stats[(size_t)b * 4 + 3] = (double)it;
    if (step_info) {
      const int fl = (any_pen ? kFemFlagPenetration : 0) | ((!accepted && !(dmax <= dx_tol)) ? kFemFlagLsFailed : 0);
      double* si = step_info + (size_t)b * 4;
      si[0] = (accumulate ? si[0] : 0.0) + 1.0;
      si[1] = dmax;
      si[2] = (double)((accumulate ? (int)si[2] : 0) | fl);
      si[3] = (accumulate ? si[3] : 0.0) + (double)it;
    }
  }
}

#include "fem_ball.h"

// ---- K17c: the same Newton iteration with the env's state resident on the CU ------------------------------------
// fem_newton_kernel above streams ~700 KB per env and PCG iteration through HBM / L2 (cached tet state, per-tet H*p rows,
// seven nodal vectors): 81 us per PCG iteration for 512 envs, bandwidth-bound.  Here
//   * thread v OWNS vertex v (V <= 512): x, r, z, p, d, H*p and the 3x3 preconditioner block live in registers;
//   * only what tets gather at random - x and p - sits in LDS (2 x 12 KB at 495 vertices);
//   * the tet state (F, cofactor, coefficients) is RECOMPUTED from the LDS x every iteration (~80 f64 FMAs) instead of
//     being re-read (96 B per tet);
//   * per-tet rows travel to their vertices through a 48 KB LDS window of 512 tets at a time; each vertex walks its CSR
//     incidence list (sorted by tet) once per iteration, so the result is deterministic (the order differs from the
//     streaming kernel's only by the tet renumbering of tacex_fem_create).
// Mesh constants (tets, DmInv, vol, CSR) are the only global reads inside the PCG loop and are shared by all envs (L2).
constexpr int kNwtThreads = 512;
constexpr int kNwtChunk = 512;  // tets per LDS exchange window (1024 = 2 per thread measured slower: 90 spilled VGPRs)
constexpr int kNwtTpw = kNwtChunk / kNwtThreads;
// MESHES OF MORE THAN 512 VERTICES run the same kernel with 768 threads (thread v still owns vertex v; 3 waves per SIMD instead of 2,
// i.e. 168 registers per lane instead of 256: more of the per-thread state goes to scratch, which is what such an env pays for
// staying on one CU - the streaming kernel's alternative is a trip through HBM per PCG iteration).  Only the atomic flavour exists
// there: without the exchange window and the incidence list in LDS (12 x NT doubles + 4 T shorts) a 593-vertex / 2 003-tet mesh with
// friction fits the CU's 160 KB.  The region between p and the reduction rows then only has to hold what the kernel parks in it: the
// 15 V doubles of D | E blocks the chain factorisation exchanges (p + region), the preconditioner's r | r_c | y_c | z.  What bounds the
// vertex count is the LDS, not the threads: 268 bytes per vertex with friction (about 600 vertices), 212 without (about 745) - a
// 1 024-thread variant would never be launched and is not instantiated.
__host__ __device__ constexpr int nwt_window_doubles(int V, int NT) {
  // (wide variants: p + this region carry the (V,15) D | E blocks the chain threads exchange for the factorisation - 12 V doubles - and,
  //  between sweeps, the preconditioner's r | r_c | y_c | z.  Shrinking it to the latter was tried in round 5: the factorisation then
  //  overwrites the reduction rows and the friction lag - a 550-vertex pad stopped yielding to its indenter.)
  return NT <= kNwtThreads ? 12 * kNwtChunk : (12 * V > 6 * V + 6 * kFemMaxCoarse ? 12 * V : 6 * V + 6 * kFemMaxCoarse);
}
// dynamic LDS of fem_newton_lds_kernel<., ., NT>: x, p | window | sums | [friction lag] | diagonal mass term (doubles) || chain factors |
// [friction Hessian blocks] (floats) || [incidence codes] | chain links | chain heads per thread | [CSR offsets] (u16)
static size_t nwt_lds_bytes(int V, int T, bool fric, int NT) {
  const bool big = NT > kNwtThreads;
  return ((((size_t)7 * V + (size_t)nwt_window_doubles(V, NT) + 2 * (NT / 64) + 2 + (fric ? (size_t)4 * V : 0)) * sizeof(double) +
           ((size_t)15 * V + (fric ? (size_t)6 * V : 0)) * sizeof(float) +
           ((big ? 0 : (size_t)4 * T + V + 1) + 2 * (size_t)V + NT) * sizeof(unsigned short)) + 15) & ~(size_t)15;
}

// block-wide sum with ONE barrier: wave partials go to one of two alternating LDS rows and every thread adds them in the
// same fixed order (the row written two calls ago cannot still be read: a barrier lies in between)
template <int NT>
__device__ __forceinline__ double block_sum1(double v, double* sh2 /* 2 x NT / 64 doubles */, int& phase) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  double* row = sh2 + (NT / 64) * (phase & 1);
  ++phase;
  if ((threadIdx.x & 63) == 0) row[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) s += row[w];
  return s;
}

template <bool MESH, int NT>
__device__ __forceinline__ double env_energy_lds(const FemDev& m, const double* xl, const double x3[3], const double* xt,
                                                 bool own, bool c, const double* aim, double* sh, int& phase,
                                                 const double* ind = nullptr, double wv = 0.0, const double* fv = nullptr,
                                                 const double* xn = nullptr, const double* disp = nullptr, const double* e_tets = nullptr) {
  // e_tets: this thread's share of the elastic energy, already summed over the same tets in the same order by the gradient sweep of the
  // Newton iteration (same x): the line search's E(x) then costs no tet sweep of its own - and is the same bits as with one
  double e = e_tets ? *e_tets : 0.0;
  const double dt2 = m.dt * m.dt;
  for (int t = threadIdx.x; t < (e_tets ? 0 : m.T); t += blockDim.x) {
    int v[4];
    double Di[9], F[9];
    double vol_t;
    load_tet_blk(m, t, v, Di, vol_t);
    deformation_gradient(xl, v, Di, F);
    TetState s;
    tet_state(m, F, s);
    e += dt2 * vol_t * psi_of(m, s);
  }
  if (own) {
    const int v = threadIdx.x;
    const double mv = m.mass[v];
    double q = 0.0, qc = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double d = x3[i] - xt[v * 3 + i];
      q += d * d;
      if (c) { const double cc = x3[i] - aim[v * 3 + i]; qc += cc * cc; }
    }
    e += 0.5 * mv * q + 0.5 * m.strength * mv * qc;
    if (ind) e += dt2 * contact_eval<MESH>(m, ind, wv, x3, false).e;
    if (fv) {
      const double xn3[3] = {xn[v * 3], xn[v * 3 + 1], xn[v * 3 + 2]};
      e += dt2 * friction_eval(m.fric_mu, m.fric_eps, fv + v * 4, x3, xn3, disp, false).e;
    }
  }
  return block_sum1<NT>(e, sh, phase);
}

// block-wide minimum, same one-barrier scheme as block_sum1
template <int NT>
__device__ __forceinline__ double block_min1(double v, double* sh2, int& phase) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  double* row = sh2 + (NT / 64) * (phase & 1);
  ++phase;
  if ((threadIdx.x & 63) == 0) row[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = row[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) s = fmin(s, row[w]);
  return s;
}

// Backtracking beyond the configured cap: when the capped search (LineSearch.max_iter, US:96-101) finds no decrease the step is
// halved further, down to 2^-kLsRescue - a Newton direction computed BEFORE a vertex enters the barrier zone knows nothing of
// the barrier it runs into (soft gel, 10 GPa contact resistance: the admissible step can be 1e-3 of the CCD bound), and an env
// whose search failed would otherwise repeat the same failing iteration until the iteration cap.
constexpr int kLsRescue = 32;
// PSD-SAFE MODE.  The Stable Neo-Hookean Hessian is applied matrix-free and unprojected: aI + b f f^T + lam c c^T + c_J d2J/dF2, whose last
// term is indefinite (eigenvalues +-sigma_k).  On the gelpad - a thin pad held on its whole back face - the sum stays positive definite
// along the directions the PCG visits; a slender body bent by its indenter (simple_axle held at its ends) has compressed elements with
// negative curvature, the PCG breaks off on its first direction, and with a coarse correction in M^-1 that direction is a soft global
// mode hundreds of metres long: the Newton loop crawls into inverted states (replayed in the oracle: 40 iterations, then 6 mm of
// "dent" on a 3 mm rod).  IPC projects every element Hessian onto the PSD cone; here an env that MEETS negative curvature (p^T H p <= 0)
// switches, for that Newton iteration (which starts over), to a Hessian in which |c_J| is clamped to a / sqrt(2 Ic) per element: the spectral norm of
// d2J/dF2 is below sqrt(2 Ic) (its eigenvalues are +-sigma_k and those of the scaling block, bounded by the sum of two singular
// values), so a I + c_J d2J/dF2 stays positive semi-definite and with it the element Hessian.  The gradient is untouched - the
// iteration becomes a quasi-Newton one on the same minimiser - and envs that never meet negative curvature never pay.
// The PCG stops on the M^-1 norm of the residual, and M^-1 contains the coarse operator of the REST state - without the barrier and
// friction stiffness of the current contacts.  Where the coarse space holds nearly free modes (a slender body held at its ends:
// simple_axle) those modes map the contact forces in b to a huge b^T M^-1 b, the relative test passes after a handful of iterations
// with the 2-norm of the residual ABOVE that of b (measured at exit: 3.7 x |b| in the median there; on the gelpad, whose back face
// is held, <= 0.004 x |b| with the test on the norm of rounds 1-4 and a few per cent with libuipc's test on r.z), and the Newton loop crawls on such directions until a line search fails.  Safeguard: if the
// residual's 2-norm at exit is above kCoarseTrust x |b| - no reduction at all -, the env drops the coarse correction for the rest of the time step (chains /
// block Jacobi alone: the test is then in a norm that sees the contact blocks) and the iteration starts over.
constexpr double kCoarseTrust = 1.0;

// One launch = up to `max_newton` Newton iterations of every env (tacex_fem_step: the whole Newton loop of world.advance(),
// US:250-252, without a host round trip; tacex_fem_newton_step: max_newton = 1).  An env leaves the loop when the Newton
// direction of an iteration moves no vertex by more than dx_tol (velocity_tol * dt, US:62-66) - the criterion looks at the UNSCALED
// direction (IPC's test on the search direction): a CCD- or search-shortened update says nothing about convergence.
// ATOM: per-tet rows are ADDED into per-vertex LDS accumulators with ds_add_f64 instead of travelling through the 48 KB exchange
// window and a CSR gather: 2 barriers per sweep instead of 8, no gather phase, and the atomics of one tet batch overlap the f64
// arithmetic of the next (no barrier between batches: wave skew no longer costs).  The price is the summation ORDER of a vertex's
// ~24 contributions, which then depends on the timing of the waves: results agree to round-off (1e-16 relative per add), not
// bit for bit from run to run.  `tacex_fem_set_deterministic(ctx, 1)` selects the window path (ATOM = false).
template <bool MESH, bool ATOM, int NT>
__global__ __launch_bounds__(NT) void fem_newton_lds_kernel(FemDev m, double* xg, const double* xtg,
                                                                     const uint8_t* consg, const double* aimg, double* stats,
                                                                     int pcg_max_iter, double pcg_tol_rate, int ls_max_iter,
                                                                     double* dxg, double dx_tol, int max_newton, double* step_info,
                                                                     const double* xprevg, const double* dispg, const int* env_order,
                                                                     int follow, double* lagg) {
  extern __shared__ __attribute__((aligned(16))) double nlds[];
  constexpr int CH = NT;  // tets per pass of a sweep (one per thread) = tets per exchange window
  constexpr bool BIG = NT > kNwtThreads;  // meshes of more than 512 vertices: see nwt_lds_bytes
  static_assert(!BIG || ATOM, "the wide variants exist in the atomic flavour only");
  const int V = m.V, T = m.T;
  const int wave_s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tid = fresh_tid(wave_s);
  double* xs = nlds;            // (V,3) current x
  double* ps = xs + 3 * V;      // (V,3) PCG direction p, later the line-search candidate
  double* hv = ps + 3 * V;      // (12, CH) per-tet rows of the current window | accumulators, exchange arrays (nwt_window_doubles)
  double* sh = hv + nwt_window_doubles(V, NT);  // 2 x NT / 64 wave partials of block_sum1 (+2 pad)
  // friction (tacex_fem_step with a friction ratio): (V,4) lagged normal force and normal, (V,6) Hessian blocks (floats)
  const bool fric_lds = m.indenters && m.area && m.fric_mu > 0.0 && xprevg != nullptr && dispg != nullptr;  // (xprevg is only handed over with friction)
  double* fl = sh + 2 * (NT / 64) + 2;
  double* mdl = fl + (fric_lds ? 4 * V : 0);  // (V) diagonal mass term m_v (1 + s c_v): read back per PCG iteration (per-thread constants
                                              // carried in registers across the tet arithmetic went to scratch)
  float* cf = reinterpret_cast<float*>(mdl + V);  // (V,15) chain factors: S^-1 (6, upper triangle) | G (9)
  float* fh = cf + 15 * V;
  unsigned short* csr = reinterpret_cast<unsigned short*>(fh + (fric_lds ? 6 * V : 0));  // (4T) incidence codes tet * 4 + local, vertex-major
  unsigned short* cnx = csr + (BIG ? 0 : 4 * T);  // (V) chain successor | (V) predecessor, 0xffff = none
  unsigned short* cpv = cnx + V;
  unsigned short* chd = cpv + V;      // (NT) head vertex of the chain thread t factors and solves, 0xffff = none
  unsigned short* vto = chd + NT;  // (V+1) CSR offsets of the incidence codes (4 T < 65535); like csr not there in the wide variants
  int phase = 0;  // block_sum1 row toggle
  // env_order: envs sorted by the solver work of their PREVIOUS time step, heaviest first (fem_env_order_kernel).  One env
  // occupies one CU for its whole Newton loop and a shard brings several envs per CU, so the launch ends with whatever the
  // last-started envs need: started in index order, the heavy envs of a scene may all come last.  Workgroups are dispatched
  // in blockIdx order, so this is longest-processing-time-first list scheduling with last step's cost as the estimate.
  const int b = env_order ? env_order[blockIdx.x] : (int)blockIdx.x;
  if (dxg && dxg[b] <= dx_tol) {  // this env's last update was below the Newton tolerance (uipc_sim.py:62-66): nothing to do
    if (threadIdx.x == 0) { stats[(size_t)b * 4 + 2] = 0.0; stats[(size_t)b * 4 + 3] = 0.0; }
    return;
  }
  const size_t o = (size_t)b * V * 3;
  double* x = xg + o;
  const double* xt = xtg + o;
  const double* aim = aimg ? aimg + o : nullptr;
  const bool own = tid < V;
  const bool c = own && consg && consg[(size_t)b * V + tid];
  const double dt2 = m.dt * m.dt;
  const int nchunk = (T + CH - 1) / CH;
  const double mv = own ? m.mass[tid] : 0.0;
  const double md = mv * (1.0 + (c ? m.strength : 0.0));
  // contact: this vertex's weight and the env's indenter (nullptr: contact off)
  const double* ind = (m.indenters && m.area) ? m.indenters + (size_t)b * 8 : nullptr;
  const double wv = (ind && own) ? m.area[tid] : 0.0;
  // friction needs the positions the step started from: tacex_fem_step only
  const bool fric = ind && m.fric_mu > 0.0 && xprevg != nullptr && dispg != nullptr;
  const double* xn = fric ? xprevg + o : nullptr;
  double disp3[3] = {0, 0, 0};
  if (ind && dispg) { disp3[0] = dispg[b * 3]; disp3[1] = dispg[b * 3 + 1]; disp3[2] = dispg[b * 3 + 2]; }

  double x3[3] = {0, 0, 0};
  if (own) {
#pragma unroll
    for (int i = 0; i < 3; ++i) x3[i] = x[tid * 3 + i];
    // CONTACT-FOLLOWING START of the Newton loop (tacex_fem_step, `follow`): a surface vertex the indenter RETREATS from (its surface
    // moves away along the vertex's contact normal: disp . n < 0) and that sat inside the barrier zone before the move starts the
    // iteration displaced by that normal component, x += (disp . n) n - back at the gap it had.  Only the initial guess changes; the
    // minimiser of the step's incremental potential is what it was.  Without it a retreating indenter leaves its contact vertices
    // outside the zone, the first Newton direction springs the dent back by a millimetre into a barrier that is a hard wall for this
    // soft gel, the line search cuts the step to 1e-3 and the contact set is rediscovered a ring of vertices per iteration: 4-5 Newton
    // iterations of ~25 PCG iterations per env and step (stragglers: 17-32 iterations, 1 000-1 800 PCG iterations on ONE CU, and the
    // launch waits for them) against 2 iterations of 2 while the indenter presses.  Where the indenter APPROACHES (disp . n >= 0)
    // nothing is moved: the shrunken gap raises the barrier force and that start already converges in two iterations (following
    // there was measured: the over-displaced surface has to come back up into the barrier and line searches fail).
    if ((follow & 1) && ind && wv > 0.0 && (disp3[0] != 0.0 || disp3[1] != 0.0 || disp3[2] != 0.0)) {
      const ContactEval c0 = contact_eval<MESH>(m, ind, wv, x3);
      const double dn = disp3[0] * c0.n[0] + disp3[1] * c0.n[1] + disp3[2] * c0.n[2];
      if (dn < 0.0 && !c0.penetrating && c0.d < 1e299) {
        const double xm[3] = {x3[0] + dn * c0.n[0], x3[1] + dn * c0.n[1], x3[2] + dn * c0.n[2]};
        const ContactEval cf = contact_eval<MESH>(m, ind, wv, xm);
        if (cf.active && !cf.penetrating) { x3[0] = xm[0]; x3[1] = xm[1]; x3[2] = xm[2]; }
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) xs[tid * 3 + i] = x3[i];
  }
  if constexpr (!BIG)
    for (int k = tid; k < 4 * T; k += NT) csr[k] = (unsigned short)m.vt_idx[k];
  if (own) {
    cnx[tid] = (unsigned short)(m.ch_next ? m.ch_next[tid] : -1);
    cpv[tid] = (unsigned short)(m.ch_prev ? m.ch_prev[tid] : -1);
  }
  const int nch = m.ch_next ? m.nch : V;
  // the chain a thread factors and solves: chains are dealt from the TOP thread down, so that their solves overlap the coarse
  // solve, which keeps the low threads busy.  The head vertex is looked up in LDS where it is needed (chain_head).
  {
    const int my_chain = NT - 1 - tid;
    chd[tid] = (unsigned short)(my_chain < nch ? (m.ch_next ? m.ch_head[my_chain] : my_chain) : 0xffff);
    if constexpr (!BIG)
      for (int k = tid; k <= V; k += NT) vto[k] = (unsigned short)m.vt_off[k];
    if (own) mdl[tid] = md;
  }
  auto chain_head = [&](int t) -> int { const int h = chd[t]; return h == 0xffff ? -1 : h; };
  __syncthreads();

#ifdef TACEX_FEM_CLOCK  // debug build: cycles (s_memtime) of the sections of a PCG iteration, group TACEX_FEM_CLOCK of four -> stats
  double fclk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double fnw[4] = {0, 0, 0, 0};  // group 3: phases of a Newton iteration [gradient, block assembly + factorisation, PCG, line search]
  long long ftk = 0;
#define FEM_TICK(k) do { const long long now_ = __builtin_readcyclecounter(); fclk[k] += (double)(now_ - ftk); ftk = now_; } while (0)
#define FEM_TICK0() do { ftk = __builtin_readcyclecounter(); } while (0)
#else
#define FEM_TICK(k) do { } while (0)
#define FEM_TICK0() do { } while (0)
#endif
  // one sweep over the tets in windows of CH = kNwtTpw x 512: `make(v, Di, vol, rows)` fills the 12 rows of a tet (each thread
  // owns kNwtTpw independent tets of the window), then vertex `tid` adds the rows of its incident tets inside the window
  // (CSR entries are sorted by tet, so a cursor suffices).  The mesh constants of the NEXT window are fetched while this
  // window is computed and gathered (they are the only global reads of the sweep).
  // pre_zeroed (ATOM only): the caller has zeroed the accumulators before a barrier of its own (saves the sweep's first barrier)
  auto sweep = [&](auto&& make, double acc[3], bool pre_zeroed = false) {
    if constexpr (ATOM) {
      const int tid_a = fresh_tid(wave_s);
      double* av = hv;  // (V,3) accumulators at the head of the (otherwise idle) window region
      if (!pre_zeroed && tid_a < V) { av[tid_a * 3] = 0.0; av[tid_a * 3 + 1] = 0.0; av[tid_a * 3 + 2] = 0.0; }
      int vn[kNwtTpw][4];
      double Din[kNwtTpw][9], voln[kNwtTpw];
#pragma unroll
      for (int u = 0; u < kNwtTpw; ++u) {
        const int t = u * NT + tid_a;
        voln[u] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) vn[u][k] = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) Din[u][k] = 0.0;
        if (t < T) load_tet_blk(m, t, vn[u], Din[u], voln[u]);
      }
      if (!pre_zeroed) __syncthreads();  // accumulators zeroed (and every reader of the window region's previous content is past it)
      for (int j = 0; j < nchunk; ++j) {
        int v[kNwtTpw][4];
        double Di[kNwtTpw][9], vol[kNwtTpw];
#pragma unroll
        for (int u = 0; u < kNwtTpw; ++u) {
          vol[u] = voln[u];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[u][k] = vn[u][k];
#pragma unroll
          for (int k = 0; k < 9; ++k) Di[u][k] = Din[u][k];
          const int tn = (j + 1) * CH + u * NT + tid_a;
          if (tn < T) load_tet_blk(m, tn, vn[u], Din[u], voln[u]);
        }
#pragma unroll
        for (int u = 0; u < kNwtTpw; ++u) {
          if (j * CH + u * NT + tid_a < T) {
            double rows[12];
            make(v[u], Di[u], vol[u], rows);
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
              for (int i = 0; i < 3; ++i) atomicAdd(&av[v[u][w] * 3 + i], rows[w * 3 + i]);
          }
        }
      }
      __syncthreads();
      acc[0] = tid_a < V ? av[tid_a * 3] : 0.0; acc[1] = tid_a < V ? av[tid_a * 3 + 1] : 0.0; acc[2] = tid_a < V ? av[tid_a * 3 + 2] : 0.0;
      return;
    }
    const int tid_s = fresh_tid(wave_s);
    int e = tid_s < V ? (int)vto[tid_s] : 0;
    const int e_end = tid_s < V ? (int)vto[tid_s + 1] : 0;
    int code = e < e_end ? (int)csr[e] : 0x7fffffff;
    int code1 = e + 1 < e_end ? (int)csr[e + 1] : 0x7fffffff;
    acc[0] = acc[1] = acc[2] = 0.0;
    int vn[kNwtTpw][4];
    double Din[kNwtTpw][9], voln[kNwtTpw];
#pragma unroll
    for (int u = 0; u < kNwtTpw; ++u) {
      const int t = u * NT + tid_s;
      voln[u] = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) vn[u][k] = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) Din[u][k] = 0.0;
      if (t < T) load_tet_blk(m, t, vn[u], Din[u], voln[u]);
    }
    for (int j = 0; j < nchunk; ++j) {
      FEM_TICK0();
      int v[kNwtTpw][4];
      double Di[kNwtTpw][9], vol[kNwtTpw];
#pragma unroll
      for (int u = 0; u < kNwtTpw; ++u) {
        vol[u] = voln[u];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = vn[u][k];
#pragma unroll
        for (int k = 0; k < 9; ++k) Di[u][k] = Din[u][k];
        const int tn = (j + 1) * CH + u * NT + tid_s;
        if (tn < T) load_tet_blk(m, tn, vn[u], Din[u], voln[u]);
      }
#pragma unroll
      for (int u = 0; u < kNwtTpw; ++u) {
        const int tl = u * NT + tid;
        if (j * CH + tl < T) {
          double rows[12];
          make(v[u], Di[u], vol[u], rows);
#pragma unroll
          for (int k = 0; k < 12; ++k) hv[k * CH + tl] = rows[k];
        }
      }
      FEM_TICK(8);
      __syncthreads();
      FEM_TICK(9);
      const int tend4 = (j + 1) * CH * 4;
      // `code` / `code1` always hold entries e and e + 1 (INT_MAX past the list).  Two entries of the window are gathered per
      // round trip where there are two - their six rows are in flight together and are added in entry order - so a vertex
      // pays one LDS latency per PAIR of incident tets (a vertex has ~4 per window)
      while (code1 < tend4) {
        const int l0 = code & 3, t0 = (code >> 2) - j * CH, l1 = code1 & 3, t1 = (code1 >> 2) - j * CH;
        e += 2;
        const int n0 = e < e_end ? (int)csr[e] : 0x7fffffff;
        const int n1 = e + 1 < e_end ? (int)csr[e + 1] : 0x7fffffff;
        const double h00 = hv[(l0 * 3 + 0) * CH + t0], h01 = hv[(l0 * 3 + 1) * CH + t0], h02 = hv[(l0 * 3 + 2) * CH + t0];
        const double h10 = hv[(l1 * 3 + 0) * CH + t1], h11 = hv[(l1 * 3 + 1) * CH + t1], h12 = hv[(l1 * 3 + 2) * CH + t1];
        acc[0] += h00; acc[1] += h01; acc[2] += h02;
        acc[0] += h10; acc[1] += h11; acc[2] += h12;
        code = n0; code1 = n1;
      }
      if (code < tend4) {
        const int l = code & 3, tl = (code >> 2) - j * CH;
        ++e;
        acc[0] += hv[(l * 3 + 0) * CH + tl];
        acc[1] += hv[(l * 3 + 1) * CH + tl];
        acc[2] += hv[(l * 3 + 2) * CH + tl];
        code = code1;
        code1 = e + 1 < e_end ? (int)csr[e + 1] : 0x7fffffff;
      }
      FEM_TICK(10);
      if (j + 1 < nchunk) __syncthreads();
      FEM_TICK(11);  // the window is rewritten; after the last one the caller's next barrier suffices
    }
  };

  int n_newton = 0, flags = 0;
  bool done = false;
  double pcg_total = 0.0, dmax_last = INFINITY;
  // warm start of the next iteration's PCG: the part of this iteration's Newton direction the CCD filter / the line search cut off
  // (the direction is parked in the env's workspace block, behind the (V,16) preconditioner blocks: it is written once and read at most
  //  once per Newton iteration - as three more live f64 registers across the PCG loop it was part of what the allocator sent to scratch)
  double* const dprev_g = lagg + (size_t)gridDim.x * 16 * V + ((size_t)b * V + (own ? tid : 0)) * 3;
  double frac_prev = 0.0;  // (1 - accepted step) of the previous iteration, 0 when it was taken in full or rejected
  // FRICTION LAG AT THE START OF THE STEP (`follow` bit 1, the default): normal force and normal are taken in the FIRST iteration, at the
  // state the step starts from - IPC's lag "from the previous time step" (Li et al. 2020, section 5.4): that state is the previous step's
  // equilibrium, whose contact reaction is the previous normal force - and friction acts from the first iteration on.  Bit clear: the lag is
  // taken where the normal-contact solve of THIS step converged, in an iteration of its own (rounds 3-4: one more Newton iteration per step
  // with contact - a pressing step is two iterations instead of one).
  const bool lag_at_start = (follow & 2) != 0;
  bool fric_phase = fric && lag_at_start;   // friction terms are on (see friction_eval)
  bool lag_pending = fric && lag_at_start;  // the friction lag is taken in this iteration (behind the gradient and the contact evaluation)
  bool use_coarse = m.nc > 0;  // (block-uniform) false once the safeguard of kCoarseTrust has fired
  bool psd_safe = false;       // (block-uniform) see kFemFlagPsdSafe
  for (int nit = 0; nit < max_newton; ++nit) {
  if (nit > 0) __syncthreads();  // xs carries the accepted candidate of the previous iteration
  bool snapped_once = false;     // the edge snap (below) restarts an iteration at most once
  // every iteration tries the exact Hessian first: near the minimiser it is what converges quadratically.  (Measured against keeping
  // the mode for the rest of the step - faster on the axle, 273 against 441 ms, but its last iterations converge linearly and stop on
  // the tolerance 50 um from the minimiser - and against keeping it until a full step is accepted: gelpad scene 3.08 against 2.91 ms.)
  psd_safe = false;
restart_iteration:
#ifdef TACEX_FEM_CLOCK
  long long fph = __builtin_readcyclecounter();
#define FEM_PHASE(k) do { const long long n_ = __builtin_readcyclecounter(); fnw[k] += (double)(n_ - fph); fph = n_; } while (0)
#else
#define FEM_PHASE(k) do { } while (0)
#endif
  // ---- nodal gradient ----
  double r3[3], d3[3] = {0, 0, 0};
  double go3[3] = {0, 0, 0};  // gradient WITHOUT the contact terms (inertia + elasticity + constraints): the reaction a contact balances
  double e_tets = 0.0;  // this thread's tets' elastic energy at x: by-product of the gradient sweep, E(x) of the line search below
  {
    double a3[3];
    sweep([&](const int* v, const double* Di, double vol, double* g) {
      double F[9], r[12];
      deformation_gradient(xs, v, Di, F);
      TetState s;
      tet_state(m, F, s);
#ifndef TACEX_FEM_SEPARATE_E0
      e_tets += dt2 * vol * psi_of(m, s);
#endif
      shape_rows(Di, r);
      element_gradient(s, r, dt2 * vol, g);
    }, a3);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double gi = 0.0;
      if (own) {
        gi = a3[i] + mv * (x3[i] - xt[tid * 3 + i]);
        if (c) gi += m.strength * mv * (x3[i] - aim[tid * 3 + i]);
      }
      r3[i] = -gi;
      go3[i] = gi;
    }
  }
  // barrier of this vertex at x: gradient b1 n, curvature b2 n n^T (the b1 * hess(d) part is negative semi-definite for a
  // convex indenter and dropped: the usual PSD projection of IPC)
  const ContactEval ce = contact_eval<MESH>(m, ind, wv, x3);
  if (ce.penetrating) flags |= kFemFlagPenetration;  // (per thread; or-reduced into step_info at the end)
  if (ce.active) {
#pragma unroll
    for (int i = 0; i < 3; ++i) r3[i] -= dt2 * ce.b1 * ce.n[i];
  }
  const double cb2 = ce.active ? dt2 * ce.b2 : 0.0;
  // FRICTION LAG (normal force lam, normal n per vertex), taken once per step at the state the normal-contact solve converged to:
  // lam = min(-dB/dd, reaction), reaction = (g_other . n) / dt^2 = the normal force that balances inertia + elasticity + constraints
  // at this vertex.  In force balance the two agree (that IS the balance); but the Newton loop stops on its step-size tolerance
  // (velocity_tol * dt = 0.5 mm, US:62-66), where a contact vertex may still sit at 0.98 d_hat - and there the 10 GPa barrier pushes
  // with 87 N on a pad whose whole reaction is below 1 N.  Lagging THAT force made the first friction iteration's direction 0.66 m
  // long and the env spend 34 Newton / 1 900 PCG iterations on one CU while the launch waited (scene step 11, env 442; replayed
  // through the oracle in tests/studies/fem_straggler_replay.py: 38 / 2 280 -> 5 / 21 with the cap).
  if (lag_pending) {  // (block-uniform; every thread writes and later reads its own four doubles: no barrier)
    lag_pending = false;
    double lam = 0.0;
    if (own) {
      double ln[3] = {ce.n[0], ce.n[1], ce.n[2]};
      if (follow & 4) {
        // IPC's lag to the letter (Li et al. 2020, section 5.4: lam^n, T^n "from the previous time step"): the barrier force and the normal
        // of the PREVIOUS configuration - the positions the step starts from against the indenter where it stood then (its row moved
        // back by the displacement since the previous step).  That configuration is the previous step's equilibrium, so this IS the
        // previous normal force; no cap, nothing of the current iterate enters (tacex_fem_set_friction_lag, mode 1).
        double indp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) indp[k] = ind[k];
        indp[1] -= disp3[0]; indp[2] -= disp3[1]; indp[3] -= disp3[2];
        const double xn3[3] = {xn[tid * 3], xn[tid * 3 + 1], xn[tid * 3 + 2]};
        const ContactEval cp = contact_eval<MESH>(m, indp, wv, xn3);
        lam = (cp.active && !cp.penetrating) ? -cp.b1 : 0.0;
        ln[0] = cp.n[0]; ln[1] = cp.n[1]; ln[2] = cp.n[2];
      } else if (ce.active) {
        const double react = (go3[0] * ce.n[0] + go3[1] * ce.n[1] + go3[2] * ce.n[2]) / dt2;
        lam = fmin(-ce.b1, fmax(react, 0.0));
      }
      const bool on = lam > 0.0;
      fl[tid * 4] = on ? lam : 0.0;
      fl[tid * 4 + 1] = on ? ln[0] : 0.0; fl[tid * 4 + 2] = on ? ln[1] : 0.0; fl[tid * 4 + 3] = on ? ln[2] : 0.0;
    }
    if (!__syncthreads_or(lam > 0.0)) {  // no vertex carries a normal force: nothing for friction to act on
      if (!lag_at_start) { done = true; break; }  // (two-phase mode: normal contact had converged, the step is done)
      fric_phase = false;                         // (lag at the start: this step runs without friction)
    }
  }
  // friction of this vertex at x: gradient into the residual, Hessian block into LDS (read back by H.p and the preconditioner)
  if (fric_phase && own) {
    const double xn3[3] = {xn[tid * 3], xn[tid * 3 + 1], xn[tid * 3 + 2]};
    const FricEval fe = friction_eval(m.fric_mu, m.fric_eps, fl + tid * 4, x3, xn3, disp3, true);
#pragma unroll
    for (int i = 0; i < 3; ++i) r3[i] -= dt2 * fe.g[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) fh[tid * 6 + k] = (float)(dt2 * fe.h[k]);
  }
  FEM_PHASE(0);
  // ---- block part of the preconditioner: block-tridiagonal LDL^T along vertex chains (tacex_fem_set_chains; a chain of one
  //      vertex = 3x3 block Jacobi).  Every vertex assembles its diagonal block D and the block E = A(v, next(v)) towards its
  //      chain successor (columns recomputed per incident tet), the blocks meet in LDS (the idle p / window region), and the
  //      thread of a chain walks it:  S_0 = D_0,  G_i = S_i^-1 E_i,  S_{i+1} = D_{i+1} - E_i^T G_i.  S^-1 (6) and G (9) stay in
  //      LDS as FLOATS: z = L^-T S^-1 L^-1 r is symmetric positive definite for any G as long as the S^-1 are, so the rounding
  //      costs preconditioner quality only (none measurable: profiles/r03_experiments.md section 9). ----
  //      The ELASTIC part of D and E is assembled in the FIRST iteration of a launch only and kept per env in the workspace (15
  //      doubles per vertex, SoA): between the Newton iterations of one time step the deformation gradients move by per cent, the
  //      blocks that change by orders of magnitude - barrier curvature, friction - are added fresh every iteration, and a
  //      preconditioner only has to stay SPD.  The assembly was 164 K of the 333 K cycles of an iteration in steady contact
  //      (section clock, profiles/r04_experiments.md), and every step with contact runs at least two iterations (normal contact,
  //      then friction).
  {
    double D[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // elastic part of D / E: assembled for ALL envs of the launch by fem_assemble_blocks_kernel ahead of this kernel, at the state the
    // launch starts from, (V,16) per env in the workspace (round 5; rounds 3-4 assembled it here in the first iteration of a launch -
    // 111 K of a pressing step's 268 K cycles, and the part of this kernel the register allocator spilled most for)
    if (own) {
      // one 128-byte record per vertex: D upper triangle (6) | E (9) | pad - ONE address (rebuilt here from a fresh thread id, never
      // carried across the PCG loop: fifteen hoisted [15][V] row addresses were 30 of the kernel's spilled registers)
      const double* q = lagg + ((size_t)b * V + (size_t)fresh_tid(wave_s)) * 16;
      D[0] = q[0]; D[1] = q[1]; D[2] = q[2]; D[4] = q[3]; D[5] = q[4]; D[8] = q[5];
      D[3] = D[1]; D[6] = D[2]; D[7] = D[5];
#pragma unroll
      for (int k = 0; k < 9; ++k) E[k] = q[6 + k];
      // the blocks that are never lagged: mass + constraint, barrier curvature, friction
      D[0] += md; D[4] += md; D[8] += md;
    }
    // EDGE SNAP: one exact 1-D minimisation per surface vertex that is about to run into the barrier zone from outside (or sits in its
    // outermost sliver), along its contact normal - a nonlinear Gauss-Seidel sweep over the stiffest degrees of freedom, taken before
    // the Newton system of the iteration is set up.  The barrier is C2 with b'' -> 0 at d_hat: the Newton system is blind to it for
    // such a vertex, its direction sends the vertex a millimetre deep into a wall that stops it within microns, the line search cuts
    // the step OF THE WHOLE MESH to a per cent and the next iteration repeats it (the apex vertex of a retreating contact crossed the
    // zone edge back and forth for 13 iterations with max |d| 0.6-0.8 mm against a tolerance of 0.5 mm: 18 Newton / 530 PCG
    // iterations for that env, the launch waiting; with the snap 4 / 160 - tests/studies/fem_straggler_replay.py).  Along n the
    // vertex's energy is  phi(t) = -(g.n) t + 1/2 (n.D n) t^2 + dt^2 kappa A b((gap - t) / d_hat)  (g = contact-free gradient, D =
    // the block above): if the elastic 1-D Newton step t_el = g.n / n.D n reaches the zone, the vertex moves to where the barrier
    // balances the force it has to carry, e = sqrt(lam d_hat / (3 kappa A)) below d_hat (b' ~ -3 e^2 near the edge, lam = g.n / dt^2),
    // never further than t_el.  The iteration then restarts from the moved state (gradient, contact, blocks from the lag).
    if (ind && !snapped_once) {  // (block-uniform)
      double tmove = 0.0;
      if (own && wv > 0.0 && !ce.penetrating && ce.d < 1e299) {
        const double gn = go3[0] * ce.n[0] + go3[1] * ce.n[1] + go3[2] * ce.n[2];
        if (gn > 0.0) {
          double nDn = 0.0;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) nDn += ce.n[i] * D[i * 3 + k] * ce.n[k];
          const double t_el = nDn > 0.0 ? gn / nDn : 0.0;
          const double e = fmin(fmax(sqrt(gn / dt2 * m.dhat / (3.0 * m.kappa * wv)), 1e-6), 1e-2);
          const double rest = ce.d - (1.0 - e) * m.dhat;  // distance to the balance depth
          if (rest > 0.0 && t_el > ce.d - m.dhat) tmove = fmin(t_el, rest);
        }
      }
      if (__syncthreads_or(tmove > 0.0)) {
        if (tmove > 0.0) {
#pragma unroll
          for (int i = 0; i < 3; ++i) { x3[i] -= tmove * ce.n[i]; xs[tid * 3 + i] = x3[i]; }
        }
        snapped_once = true;
        __syncthreads();
        goto restart_iteration;
      }
    }
    if (own) {
      if (fric_phase) {
        const float* h = fh + tid * 6;
        D[0] += h[0]; D[1] += h[1]; D[2] += h[2]; D[3] += h[1]; D[4] += h[3]; D[5] += h[4]; D[6] += h[2]; D[7] += h[4]; D[8] += h[5];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) D[i * 3 + k] += cb2 * ce.n[i] * ce.n[k];
    }
    double* xch = ps;  // (V,15) D (upper triangle) | E: p is idle until the PCG starts, the window once the gradient is gathered
    __syncthreads();   // every vertex has gathered the last window of the gradient sweep
    if (own) {
      double* q = xch + tid * 15;
      q[0] = D[0]; q[1] = D[1]; q[2] = D[2]; q[3] = D[4]; q[4] = D[5]; q[5] = D[8];
#pragma unroll
      for (int k = 0; k < 9; ++k) q[6 + k] = E[k];
    }
    __syncthreads();
    const int my_head = chain_head(tid);
    if (my_head >= 0) {
      int v = my_head;
      double S[9];
      {
        const double* q = xch + v * 15;
        S[0] = q[0]; S[1] = q[1]; S[2] = q[2]; S[3] = q[1]; S[4] = q[3]; S[5] = q[4]; S[6] = q[2]; S[7] = q[4]; S[8] = q[5];
      }
      while (true) {
        double Si[9];
        if (!inv3_spd(S, Si)) {  // cannot happen in exact arithmetic (PSD-projected element Hessians + mass); keep the operator SPD
          const double dm = fmax(S[0], fmax(S[4], S[8]));
          const double im = 1.0 / (dm > 0.0 ? dm : 1.0);
          Si[0] = im; Si[1] = 0; Si[2] = 0; Si[3] = 0; Si[4] = im; Si[5] = 0; Si[6] = 0; Si[7] = 0; Si[8] = im;
        }
        float* f = cf + v * 15;
        f[0] = (float)Si[0]; f[1] = (float)Si[1]; f[2] = (float)Si[2]; f[3] = (float)Si[4]; f[4] = (float)Si[5]; f[5] = (float)Si[8];
        const int n = cnx[v] == 0xffff ? -1 : (int)cnx[v];
        if (n < 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) f[6 + k] = 0.0f;
          break;
        }
        const double* Ev = xch + v * 15 + 6;
        double G[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) G[i * 3 + k] = Si[i * 3 + 0] * Ev[k] + Si[i * 3 + 1] * Ev[3 + k] + Si[i * 3 + 2] * Ev[6 + k];
#pragma unroll
        for (int k = 0; k < 9; ++k) f[6 + k] = (float)G[k];
        const double* q = xch + n * 15;  // S_next = D_next - E^T G
        const double Dn[9] = {q[0], q[1], q[2], q[1], q[3], q[4], q[2], q[4], q[5]};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) S[i * 3 + k] = Dn[i * 3 + k] - (Ev[i] * G[k] + Ev[3 + i] * G[3 + k] + Ev[6 + i] * G[6 + k]);
        v = n;
      }
    }
    __syncthreads();  // factors complete; the exchange region goes back to the PCG
  }
  // ---- preconditioner: z = D^-1 r (3x3 block Jacobi) + P A_c^-1 P^T r (additive coarse-grid correction) ----
  // Block Jacobi alone needs 120-330 PCG iterations on the thin, nearly incompressible pad: the error it cannot reach is
  // smooth over many elements.  The coarse space - trilinear hats of a small grid over the mesh, <= 64 nodes - carries those
  // modes; its operator is the Galerkin product with the REST-state matrix (constant per mesh and constraint set: factored on
  // the host once, profiles/r03_experiments.md section 4 shows it works as well as the current-state product).  Restriction is
  // a gather per coarse dof over its support, split over G threads with the partial sums added in a fixed order (deterministic,
  // no atomics); the sweep's LDS window is idle between two sweeps and carries r, the partial sums and the coarse vectors.
  const int nc3 = 3 * m.nc;
  // Gn lanes per coarse NODE for the restriction, H lanes per coarse dof for the coarse solve: powers of two that divide a wave, so
  // the partial sums of a node / dof sit in ONE wave and are added by a butterfly of lane exchanges - no LDS round trip, no barrier
  // (a fixed tree: deterministic)
  auto pow2_le = [](int v) { int p = 1; while (2 * p <= v && 2 * p <= 64) p *= 2; return p; };
  const int Gn = m.nc > 0 ? pow2_le(NT / m.nc) : 1;
  const int H = nc3 > 0 ? pow2_le(NT / nc3) : 1;
  const int Q = nc3 > 0 ? (nc3 + H - 1) / H : 0;
  auto apply_prec = [&](const double (&r)[3], double (&z)[3]) {
    double* rs = hv;                        // (V,3) residual
    double* rc = hv + 3 * V;                // (3 nc) restricted residual
    double* yc = rc + 3 * kFemMaxCoarse;    // (3 nc) coarse correction
    double* zs = yc + 3 * kFemMaxCoarse;    // (V,3) chain solve: y on the way down, z on the way back
    z[0] = z[1] = z[2] = 0.0;
    FEM_TICK(2);
    __syncthreads();  // every thread is done with the window of the last sweep
    const int tid_p = fresh_tid(wave_s);  // (fresh_tid: the LDS / table offsets below are rebuilt per application, never carried)
    const bool own_p = tid_p < V;
    if (own_p) {
#pragma unroll
      for (int i = 0; i < 3; ++i) rs[tid_p * 3 + i] = r[i];
    }
    __syncthreads();
    FEM_TICK(3);
    // chain solve z = L^-T S^-1 L^-1 r by the chain's thread: down the chain y_i = r_i - G_{i-1}^T y_{i-1}, back up
    // z_i = S_i^-1 y_i - G_i z_{i+1}
    auto chain_solve = [&]() {
      const int my_head = chain_head(tid_p);
      if (my_head < 0) return;
      int v = my_head, last = my_head;
      double y[3] = {rs[v * 3], rs[v * 3 + 1], rs[v * 3 + 2]};
      while (true) {
        zs[v * 3] = y[0]; zs[v * 3 + 1] = y[1]; zs[v * 3 + 2] = y[2];
        last = v;
        const int n = cnx[v] == 0xffff ? -1 : (int)cnx[v];
        if (n < 0) break;
        const float* g = cf + v * 15 + 6;
        const double y0 = y[0], y1 = y[1], y2 = y[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) y[k] = rs[n * 3 + k] - ((double)g[k] * y0 + (double)g[3 + k] * y1 + (double)g[6 + k] * y2);
        v = n;
      }
      v = last;
      double zn[3] = {0, 0, 0};
      while (true) {
        const float* f = cf + v * 15;
        const double y0 = zs[v * 3], y1 = zs[v * 3 + 1], y2 = zs[v * 3 + 2];
        double zz[3];
        zz[0] = (double)f[0] * y0 + (double)f[1] * y1 + (double)f[2] * y2;
        zz[1] = (double)f[1] * y0 + (double)f[3] * y1 + (double)f[4] * y2;
        zz[2] = (double)f[2] * y0 + (double)f[4] * y1 + (double)f[5] * y2;
#pragma unroll
        for (int i = 0; i < 3; ++i) zz[i] -= (double)f[6 + i * 3] * zn[0] + (double)f[7 + i * 3] * zn[1] + (double)f[8 + i * 3] * zn[2];
        zs[v * 3] = zz[0]; zs[v * 3 + 1] = zz[1]; zs[v * 3 + 2] = zz[2];
        zn[0] = zz[0]; zn[1] = zz[1]; zn[2] = zz[2];
        const int pv = cpv[v] == 0xffff ? -1 : (int)cpv[v];
        if (pv < 0) break;
        v = pv;
      }
    };
    if (nc3 == 0 || !use_coarse) {
      chain_solve();
      __syncthreads();
      if (own_p) {
#pragma unroll
        for (int i = 0; i < 3; ++i) z[i] = zs[tid_p * 3 + i];
      }
      return;
    }
    {  // Gn lanes per coarse NODE, all three components: one (vertex, weight) fetch serves three sums
      const int node = tid_p / Gn, j = tid_p - node * Gn;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      if (node < m.nc) {
        const int e1 = ldg_off<int>(m.cn_off, (unsigned)(node + 1) * 4u);
        for (int e = ldg_off<int>(m.cn_off, (unsigned)node * 4u) + j; e < e1; e += Gn) {
          const int v0 = ldg_off<int>(m.cn_vtx, (unsigned)e * 4u);
          const double w0 = ldg_off<double>(m.cn_w, (unsigned)e * 8u);
          a0 += w0 * rs[v0 * 3]; a1 += w0 * rs[v0 * 3 + 1]; a2 += w0 * rs[v0 * 3 + 2];
        }
      }
      for (int o = Gn >> 1; o > 0; o >>= 1) {  // (every lane of the wave takes part in the exchange)
        a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64);
      }
      if (node < m.nc && j == 0) { rc[node * 3] = a0; rc[node * 3 + 1] = a1; rc[node * 3 + 2] = a2; }
    }
    __syncthreads();
    FEM_TICK(5);
    {  // coarse solve y = A_c^-1 r_c: H lanes per row, each over a slice of its (contiguous) row.  The chain solves run beside it
       // on the threads it leaves idle (chains are dealt from the top).
      const int dof = tid_p / H, h = tid_p - dof * H;
      double acc = 0.0;
      if (dof < nc3) {
        const int q1 = min(nc3, (h + 1) * Q);
        const unsigned row = (unsigned)(dof * nc3) * 8u;  // (3 nc)^2 doubles <= 288 KB: 32-bit byte offsets
        for (int q = h * Q; q < q1; ++q) acc += ldg_off<double>(m.ac_inv, row + (unsigned)q * 8u) * rc[q];
      }
      for (int o = H >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (dof < nc3 && h == 0) yc[dof] = acc;
    }
    chain_solve();
    __syncthreads();
    FEM_TICK(6);
    if (own_p) {
#pragma unroll
      for (int i = 0; i < 3; ++i) z[i] = zs[tid_p * 3 + i];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int node = ldg_off<int>(m.cv_node, (unsigned)tid_p * 32u + (unsigned)k * 4u);
        const double w = ldg_off<double>(m.cv_w, (unsigned)tid_p * 64u + (unsigned)k * 8u);
#pragma unroll
        for (int i = 0; i < 3; ++i) z[i] += w * yc[node * 3 + i];
      }
    }
    FEM_TICK(7);
  };
  FEM_PHASE(1);
  // ---- PCG ----
  // Stops when the preconditioned residual has dropped to tol_rate times that of the right-hand side (r^T M^-1 r against
  // b^T M^-1 b; the same test as before for a zero start).  WARM START: when the previous iteration's step was cut short (CCD
  // bound, backtracking at the edge of the barrier zone - the release regime of a retreating indenter takes up to the iteration
  // cap of such steps), the new system differs from the old one only around the newly pinned vertices, and the unfinished part
  // (1 - step) d_prev is a far better start than zero: one extra H.d sweep buys most of the iterations.
  double z3[3], p3[3];
  double part = 0.0;
  apply_prec(r3, z3);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    p3[i] = z3[i];
    part += r3[i] * z3[i];
  }
  const double rz_b = block_sum1<NT>(part, sh, phase);
  double bb = 0.0;  // |b|^2, for the safeguard of kCoarseTrust
  if (use_coarse) bb = block_sum1<NT>(r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2], sh, phase);
  double rz = rz_b;
  bool warm = frac_prev > 0.0 && rz_b > 0.0;
  const bool warm_used = warm;
  if (warm) {
#pragma unroll
    for (int i = 0; i < 3; ++i) d3[i] = own ? frac_prev * dprev_g[i] : 0.0;
  }
  int it = 0;
  bool neg_curv = false;
  // Stopping test as libuipc's linear_pcg runs it (src/backends/cuda/linear_system/linear_pcg.cu, LinearPCG::pcg: `abs(rz_new) <= global_tol_rate *
  // rz0`, global_tol_rate = linear_system/tol_rate, US:90): relative on r^T M^-1 r ITSELF, not on its square root - tol_rate 1e-3 is a factor 0.032
  // on the M^-1 norm of the residual.  (Rounds 1-4 tested the norm, tol_rate^2 on r.z: a thousand times stricter than the reference's own
  // solver and about twice the PCG iterations.)
  while (warm || (it < pcg_max_iter && rz_b > 0.0 && rz > pcg_tol_rate * rz_b)) {
    double q3[3] = {0, 0, 0};  // the vector H is applied to (d0 of the warm start, else p): lives in ps during the sweep
    const int tid = fresh_tid(wave_s);  // (shadows the kernel-wide copy: nothing derived from it crosses an iteration)
    FEM_TICK0();
    if (own) {
#pragma unroll
      for (int i = 0; i < 3; ++i) ps[tid * 3 + i] = warm ? d3[i] : p3[i];
      if constexpr (ATOM) { hv[tid * 3] = 0.0; hv[tid * 3 + 1] = 0.0; hv[tid * 3 + 2] = 0.0; }  // the sweep's accumulators (a block reduction's
                                                                                            // barrier lies behind the last reader of the region)
    }
    __syncthreads();
    // Hq = (M + s Mc + dt^2 K) q, matrix-free: per-tet dP[dF(q)] rows, gathered per vertex
    double a3[3];
    sweep([&](const int* v, const double* Di, double vol, double* rows) {
      double F[9], dF[9], dP[9], r[12];
      deformation_gradient(xs, v, Di, F);
      deformation_gradient(ps, v, Di, dF);  // linear in q
      TetState s;
      tet_state(m, F, s);
      if (psd_safe) {
        const double lim = s.a / sqrt(2.0 * fmax(s.Ic, 1e-300));
        s.c = fmin(fmax(s.c, -lim), lim);
      }
      apply_dP(m, s, dF, dP);
      shape_rows(Di, r);
      const double sc = dt2 * vol;
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          rows[w * 3 + i] = sc * (dP[i * 3 + 0] * r[w * 3 + 0] + dP[i * 3 + 1] * r[w * 3 + 1] + dP[i * 3 + 2] * r[w * 3 + 2]);
    }, a3, ATOM);
    FEM_TICK(0);
    // (q is read back from LDS - the sweep leaves ps alone - rather than carried in registers across the tet arithmetic: the loop
    //  body holds ~250 live registers there, and every value carried across it went to scratch)
    double mdv = 0.0;
    if (own) {
#pragma unroll
      for (int i = 0; i < 3; ++i) q3[i] = ps[tid * 3 + i];
      mdv = mdl[tid];
    }
    double Hp3[3];
    part = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Hp3[i] = own ? a3[i] + mdv * q3[i] + cb2 * ce.n[i] * (ce.n[0] * q3[0] + ce.n[1] * q3[1] + ce.n[2] * q3[2]) : 0.0;
    }
    if (fric_phase && own) {
      const float* h = fh + tid * 6;
      Hp3[0] += h[0] * q3[0] + h[1] * q3[1] + h[2] * q3[2];
      Hp3[1] += h[1] * q3[0] + h[3] * q3[1] + h[4] * q3[2];
      Hp3[2] += h[2] * q3[0] + h[4] * q3[1] + h[5] * q3[2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) part += q3[i] * Hp3[i];
    if (warm) {  // r = b - H d0, then the usual start from there (not counted as an iteration)
#pragma unroll
      for (int i = 0; i < 3; ++i) r3[i] -= Hp3[i];
      apply_prec(r3, z3);
      part = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        p3[i] = z3[i];
        part += r3[i] * z3[i];
      }
      rz = block_sum1<NT>(part, sh, phase);
      warm = false;
      continue;
    }
    const double pHp = block_sum1<NT>(part, sh, phase);
    FEM_TICK(1);
    if (!(pHp > 0.0)) {  // negative curvature (block-uniform)
      if (!psd_safe) {  // PSD-safe Hessian for this iteration, which starts over (same warm start)
        psd_safe = true;
        flags |= kFemFlagPsdSafe;
        pcg_total += (double)it;  // (the work was done)
        __syncthreads();
        goto restart_iteration;
      }
      if (it == 0 && frac_prev > 0.0) {  // a warm start alone is no descent direction: once more from a zero start
        frac_prev = 0.0;
        __syncthreads();
        goto restart_iteration;
      }
      // keep d (first iteration from a zero start: preconditioned steepest descent)
      if (it == 0) { d3[0] = z3[0]; d3[1] = z3[1]; d3[2] = z3[2]; }
      neg_curv = true;
      break;
    }
    const double al = rz / pHp;
    part = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      d3[i] += al * p3[i];
      r3[i] -= al * Hp3[i];
    }
    apply_prec(r3, z3);
#pragma unroll
    for (int i = 0; i < 3; ++i) part += r3[i] * z3[i];
    const double rz_new = block_sum1<NT>(part, sh, phase);
    const double beta = rz_new / rz;
#pragma unroll
    for (int i = 0; i < 3; ++i) p3[i] = z3[i] + beta * p3[i];
    rz = rz_new;
    ++it;
  }
  if (use_coarse && !neg_curv) {  // (block-uniform)
    const double rr = block_sum1<NT>(r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2], sh, phase);
    if (rr > kCoarseTrust * kCoarseTrust * bb) {
      use_coarse = false;
      flags |= kFemFlagCoarseOff;
      pcg_total += (double)it;  // (the work was done)
      __syncthreads();
      goto restart_iteration;
    }
  }
  FEM_PHASE(2);
  // ---- backtracking line search on the incremental potential (accept the first E(x + step d) <= E(x)) ----
#ifdef TACEX_FEM_SEPARATE_E0  // (A/B hook: E(x) by a tet sweep of its own, as before)
  const double E0 = env_energy_lds<MESH, NT>(m, xs, x3, xt, own, c, aim, sh, phase, ind, wv, fric_phase ? fl : nullptr, xn, disp3);
#else
  const double E0 = env_energy_lds<MESH, NT>(m, xs, x3, xt, own, c, aim, sh, phase, ind, wv, fric_phase ? fl : nullptr, xn, disp3, &e_tets);
#endif
  double step = 1.0, E1 = E0;
  if (ind) {
    // CCD step filter for analytic indenters: a signed distance field is 1-Lipschitz, so a vertex at gap d moving by
    // step |dx| keeps a positive gap while step < d / |dx|; the largest step every surface vertex allows, with slack
    double amax = 1.0;
    if (wv > 0.0 && !ce.penetrating && ce.d < 1e299) {
      const double nd = sqrt(d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2]);
      if (nd > 0.0) amax = fmin(1.0, kCcdSlack * ce.d / nd);
    }
    step = block_min1<NT>(amax, sh, phase);
  }
  // max |d| of the UNSCALED Newton direction (what the convergence test looks at)
  const double dmax = -block_min1<NT>(own ? -fmax(fabs(d3[0]), fmax(fabs(d3[1]), fabs(d3[2]))) : 0.0, sh, phase);
  // no search starts with a vertex moving further than the body is long: a steepest-descent direction kept on negative curvature has
  // no length scale (M^-1 b with a soft coarse mode: 600 m on the 26 mm axle, out of reach of the 2^-40 the backtracking can do)
  if (dmax > m.step_cap) step = fmin(step, m.step_cap / dmax);
  const double step0 = step;  // after the CCD filter
  bool accepted = false;
  double xc3[3] = {0, 0, 0};
  const int ls_cap = ls_max_iter > kLsRescue ? ls_max_iter : kLsRescue;
  for (int ls = 0; ls <= ls_cap; ++ls) {
    __syncthreads();  // every tet is done reading ps (PCG sweep or the previous candidate)
    if (own) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { xc3[i] = x3[i] + step * d3[i]; ps[tid * 3 + i] = xc3[i]; }
    }
    __syncthreads();
    const double Ec = env_energy_lds<MESH, NT>(m, ps, xc3, xt, own, c, aim, sh, phase, ind, wv, fric_phase ? fl : nullptr, xn, disp3);
    if (Ec <= E0) { E1 = Ec; accepted = true; break; }
    step *= 0.5;
  }
  if (accepted) {
    if (own) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { x3[i] = xc3[i]; xs[tid * 3 + i] = xc3[i]; }
    }
  } else {
    step = 0.0;
    // (no decrease to be had at a point whose Newton step is below the tolerance: converged; a direction that came from a warm start
    //  has no descent guarantee: that iteration is repeated from a zero start, see below)
    if (!(dmax <= dx_tol) && !warm_used) flags |= kFemFlagLsFailed;
  }
  // A rejected search leaves x where it was and cuts the warm start: the next iteration would be bit-identical to this one, and
  // the one after, up to max_newton - each a full assembly, a PCG solve and up to 33 energy sweeps on a CU other envs wait for.
  // The env stops here with the flag set (block-uniform: `accepted` and `dmax` are block reductions).
  const bool ls_dead = !accepted && !(dmax <= dx_tol);
  FEM_PHASE(3);
  ++n_newton;
  pcg_total += (double)it;
  dmax_last = dmax;
  frac_prev = (accepted && step < 1.0) ? 1.0 - step : 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (own) dprev_g[i] = d3[i];
  if (tid == 0) {
    stats[(size_t)b * 4 + 0] = E0; stats[(size_t)b * 4 + 1] = E1; stats[(size_t)b * 4 + 2] = step; stats[(size_t)b * 4 + 3] = (double)it;
#ifdef TACEX_FEM_CLOCK
    for (int k = 0; k < 4; ++k) stats[(size_t)b * 4 + k] = (TACEX_FEM_CLOCK) == 3 ? fnw[k] : fclk[4 * ((TACEX_FEM_CLOCK) % 3) + k];
#endif
  }
  // IPC's test (Li et al. 2020, Algorithm 1: the infinity norm of the SEARCH DIRECTION over dt against the velocity tolerance): the
  // unscaled Newton direction, whatever the CCD bound and the line search then made of the step - a shortened UPDATE says nothing
  // about convergence, a short DIRECTION does.  (Rounds 2-3 also demanded a full-length accepted step: at the edge of
  // the barrier zone that never happens and every retreat step ran to the iteration cap with directions 5x below the tolerance.)
  if (ls_dead) {
    if (warm_used) continue;  // (frac_prev is 0 now: the next iteration solves the same system from a zero start)
    break;
  }
  const bool converged = dmax <= dx_tol;
  if (converged) {  // wave-uniform: every quantity above is a block reduction
    if (fric && !fric_phase && !lag_at_start) {
      // (two-phase mode) normal contact is balanced: freeze the friction lag (normal force, normal) at this state and go on, unless
      // no vertex of the env is in contact
      fric_phase = true;
      bool touching = false;
      if (own) {
        const ContactEval cf = contact_eval<MESH>(m, ind, wv, x3);
        touching = cf.active;
        fl[tid * 4] = 0.0; fl[tid * 4 + 1] = 0.0; fl[tid * 4 + 2] = 0.0; fl[tid * 4 + 3] = 0.0;
      }
      // the lag itself needs the contact-free gradient at this state: the next iteration computes it first (lag_pending)
      if (__syncthreads_or(touching)) { frac_prev = 0.0; lag_pending = true; continue; }
    }
    done = true;
    break;
  }
  }  // Newton loop
  if (own) {
#pragma unroll
    for (int i = 0; i < 3; ++i) x[tid * 3 + i] = x3[i];
  }
  // what the next launch (tacex_fem_newton_step called in a loop) reads to skip this env: the direction's max |d| once the env
  // has converged (<= dx_tol), a value above the tolerance while it has not (a shortened update must not count as convergence)
  if (dxg && tid == 0) dxg[b] = dmax_last;
  if (step_info) {
    // (__syncthreads_or returns a truth value, not the OR of the bits: one reduction per flag)
    const int any = (__syncthreads_or(flags & kFemFlagPenetration) ? kFemFlagPenetration : 0) |
                    (__syncthreads_or(flags & kFemFlagLsFailed) ? kFemFlagLsFailed : 0) |
                    (flags & (kFemFlagCoarseOff | kFemFlagPsdSafe));  // (these two are block-uniform)
    if (tid == 0) {
      step_info[(size_t)b * 4 + 0] = (double)n_newton; step_info[(size_t)b * 4 + 1] = dmax_last;
      step_info[(size_t)b * 4 + 2] = (double)any; step_info[(size_t)b * 4 + 3] = pcg_total;
    }
  }
}

// ---- elastic preconditioner blocks of every env, ahead of the Newton launch (round 5) ------------------------------------------
// Per vertex the 3x3 diagonal block D (upper triangle, 6) of dt^2 K at the state the launch starts from and the block E (9) towards
// its chain successor (tacex_fem_set_chains): (V,16) doubles per env in the workspace, read by fem_newton_lds_kernel in every
// Newton iteration of the launch (the blocks that change by orders of magnitude between iterations - barrier curvature, friction -
// are added there, fresh).  One workgroup per env, x and the (V,15) accumulators in LDS (71 KB at 495 vertices: two envs per CU, a
// 512-env shard in one round); rounds 3-4 ran this inside the first Newton iteration of fem_newton_lds_kernel.
//   ATOM:  tet-centric - every tet's state is computed once, its shares of the four diagonal blocks and of the chain blocks are
//          added with ds_add_f64 (summation order depends on wave timing: round-off level run-to-run differences);
//   !ATOM: vertex-centric over the incidence list in a FIXED order (tacex_fem_set_deterministic) - the tet state is recomputed per
//          incident vertex, four times the arithmetic, bit-identical runs.
// dxg / dx_tol: envs that converged in an earlier launch of the time step are skipped (same protocol as the Newton kernels).
// Blocks of the element Hessian in closed form.  With dF = e_k (x) r_B (row k of dF = r_B) contracted against r_A, the 9x9 Hessian
// of the Stable Neo-Hookean density (apply_dP: a dF + b (F:dF) F + lam (C:dF) C + c dC[dF]) gives the 3x3 block
//     B(A, B)[i][k] = a (r_A . r_B) delta_ik + b u_A[i] u_B[k] + lam w_A[i] w_B[k] + c eps_ikn g[n],
//     u = F r,  w = C r (C = cofactor matrix),  g = F (r_A x r_B)
// (the last term is d2J/dF2 = eps eps F contracted with r_A, r_B: antisymmetric, zero for A = B).  A diagonal block costs two
// matrix-vector products and six entries of three FMAs instead of three apply_dP calls (~100 f64 operations each) and their
// contractions: the assembly kernel went from 70 to 43 us per 512 envs (profiles/r05_experiments.md section 8); verified against the
// oracle's dpk1 to 1e-16 relative.
struct TetBlocks {
  double u[4][3], w[4][3], n2[4];
};
__device__ __forceinline__ void tet_blocks(const TetState& s, const double r[12], TetBlocks& tb) {
#pragma unroll
  for (int l = 0; l < 4; ++l) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      tb.u[l][i] = s.F[i * 3 + 0] * r[l * 3 + 0] + s.F[i * 3 + 1] * r[l * 3 + 1] + s.F[i * 3 + 2] * r[l * 3 + 2];
      tb.w[l][i] = s.C[i * 3 + 0] * r[l * 3 + 0] + s.C[i * 3 + 1] * r[l * 3 + 1] + s.C[i * 3 + 2] * r[l * 3 + 2];
    }
    tb.n2[l] = r[l * 3 + 0] * r[l * 3 + 0] + r[l * 3 + 1] * r[l * 3 + 1] + r[l * 3 + 2] * r[l * 3 + 2];
  }
}
// Row `l2` of a 4-row table with a RUNTIME l2 as an exact blend (weights 1.0 / 0.0) of constant-indexed reads: a runtime index sends
// the array to scratch, and a chain of selects is folded back into one by the optimiser (select of loads -> load of a selected address).
#define TB_SEL(arr, l2, i) (((l2) == 0 ? 1.0 : 0.0) * arr[0][i] + ((l2) == 1 ? 1.0 : 0.0) * arr[1][i] + ((l2) == 2 ? 1.0 : 0.0) * arr[2][i] + \
                            ((l2) == 3 ? 1.0 : 0.0) * arr[3][i])
#define R_SEL(r, l2, j) (((l2) == 0 ? 1.0 : 0.0) * r[j] + ((l2) == 1 ? 1.0 : 0.0) * r[3 + (j)] + ((l2) == 2 ? 1.0 : 0.0) * r[6 + (j)] + \
                         ((l2) == 3 ? 1.0 : 0.0) * r[9 + (j)])
// the off-diagonal block (vertex l, vertex l2) of the tet: E[i * 3 + k]
__device__ __forceinline__ void tet_block_offdiag(const FemDev& m, const TetState& s, const double r[12], const TetBlocks& tb, const double (&ul)[3],
                                                   const double (&wl)[3], const double (&rl)[3], int l2, double E[9]) {
  const double u2[3] = {TB_SEL(tb.u, l2, 0), TB_SEL(tb.u, l2, 1), TB_SEL(tb.u, l2, 2)};
  const double w2[3] = {TB_SEL(tb.w, l2, 0), TB_SEL(tb.w, l2, 1), TB_SEL(tb.w, l2, 2)};
  const double r2[3] = {R_SEL(r, l2, 0), R_SEL(r, l2, 1), R_SEL(r, l2, 2)};
  const double dot = rl[0] * r2[0] + rl[1] * r2[1] + rl[2] * r2[2];
  const double x3[3] = {rl[1] * r2[2] - rl[2] * r2[1], rl[2] * r2[0] - rl[0] * r2[2], rl[0] * r2[1] - rl[1] * r2[0]};
  double g[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) g[n] = s.c * (s.F[n * 3 + 0] * x3[0] + s.F[n * 3 + 1] * x3[1] + s.F[n * 3 + 2] * x3[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) E[i * 3 + k] = s.b * ul[i] * u2[k] + m.lam * wl[i] * w2[k];
  const double ad = s.a * dot;
  E[0] += ad; E[4] += ad; E[8] += ad;
  E[1] += g[2]; E[2] -= g[1]; E[3] -= g[2]; E[5] += g[0]; E[6] += g[1]; E[7] -= g[0];
}

#ifdef TACEX_FEM_ASM_NOATOM  // timing probe only (racy, wrong sums): what the LDS atomics of the tet-centric assembly cost
#define ASM_ADD(p, v) (*(p) += (v))
#else
#define ASM_ADD(p, v) atomicAdd((p), (v))
#endif
template <bool ATOM>
__global__ __launch_bounds__(512) void fem_assemble_blocks_kernel(FemDev m, const double* __restrict__ xg, double* __restrict__ lagg,
                                                                   const double* __restrict__ dxg, double dx_tol) {
  extern __shared__ __attribute__((aligned(16))) double alds[];
  constexpr int NT = 512;
  const int V = m.V, T = m.T, b = blockIdx.x, tid = threadIdx.x;
  if (dxg && dxg[b] <= dx_tol) return;
  double* xs = alds;           // (V,3)
  double* xa = xs + 3 * V;     // (V,15) accumulators (ATOM)
  const double* x = xg + (size_t)b * V * 3;
  double* lagw = lagg + (size_t)b * 16 * V;  // (V,16): D upper triangle (6) | E (9) | pad, one 128-byte record per vertex
  for (int k = tid; k < 3 * V; k += NT) xs[k] = x[k];
  if constexpr (ATOM)
    for (int k = tid; k < 15 * V; k += NT) xa[k] = 0.0;
  __syncthreads();
  const double dt2 = m.dt * m.dt;
  if constexpr (ATOM) {
    for (int t = tid; t < T; t += NT) {
      int v[4];
      double Di[9], F[9], r[12], vol_t;
      load_tet_blk(m, t, v, Di, vol_t);
      deformation_gradient(xs, v, Di, F);
      TetState s;
      tet_state(m, F, s);
      shape_rows(Di, r);
      TetBlocks tb;
      tet_blocks(s, r, tb);
      const double sc = dt2 * vol_t;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        double* q = xa + v[l] * 15;
        const double ul[3] = {tb.u[l][0], tb.u[l][1], tb.u[l][2]}, wl[3] = {tb.w[l][0], tb.w[l][1], tb.w[l][2]};
        const double an = s.a * tb.n2[l];
        // upper triangle: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2) -> q[0..5]
        ASM_ADD(&q[0], sc * (an + s.b * ul[0] * ul[0] + m.lam * wl[0] * wl[0]));
        ASM_ADD(&q[1], sc * (s.b * ul[0] * ul[1] + m.lam * wl[0] * wl[1]));
        ASM_ADD(&q[2], sc * (s.b * ul[0] * ul[2] + m.lam * wl[0] * wl[2]));
        ASM_ADD(&q[3], sc * (an + s.b * ul[1] * ul[1] + m.lam * wl[1] * wl[1]));
        ASM_ADD(&q[4], sc * (s.b * ul[1] * ul[2] + m.lam * wl[1] * wl[2]));
        ASM_ADD(&q[5], sc * (an + s.b * ul[2] * ul[2] + m.lam * wl[2] * wl[2]));
        const int nv = m.ch_next ? m.ch_next[v[l]] : -1;
        const int l2 = nv < 0 ? -1 : (v[0] == nv ? 0 : (v[1] == nv ? 1 : (v[2] == nv ? 2 : (v[3] == nv ? 3 : -1))));
        if (l2 >= 0) {  // this tet also holds the chain successor of vertex l: its share of the block (v_l, next(v_l))
          const double rl[3] = {r[l * 3 + 0], r[l * 3 + 1], r[l * 3 + 2]};
          double E[9];
          tet_block_offdiag(m, s, r, tb, ul, wl, rl, l2, E);
#pragma unroll
          for (int k = 0; k < 9; ++k) ASM_ADD(&q[6 + k], sc * E[k]);
        }
      }
    }
    __syncthreads();
    for (int k = tid; k < 16 * V; k += NT) {  // (V,15) -> (V,16): coalesced stores
      const int vv = k >> 4, j = k & 15;
      lagw[k] = j < 15 ? xa[vv * 15 + j] : 0.0;
    }
  } else {
    for (int vtx = tid; vtx < V; vtx += NT) {
      double D[6] = {0, 0, 0, 0, 0, 0};
      double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      const int nv = m.ch_next ? m.ch_next[vtx] : -1;
      for (int e = m.vt_off[vtx], e_end = m.vt_off[vtx + 1]; e < e_end; ++e) {
        const int code = m.vt_idx[e];
        const int t = code >> 2, l = code & 3;
        int v[4];
        double Di[9], F[9], r[12], vol_t;
        load_tet_rec(m, t, v, Di, vol_t);  // (vertex order: every lane another tet - the AoS record, not 14 scattered SoA loads)
        deformation_gradient(xs, v, Di, F);
        TetState s;
        tet_state(m, F, s);
        shape_rows(Di, r);
        const double sc = dt2 * vol_t;
        const double rl[3] = {R_SEL(r, l, 0), R_SEL(r, l, 1), R_SEL(r, l, 2)};
        double ul[3], wl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          ul[i] = s.F[i * 3 + 0] * rl[0] + s.F[i * 3 + 1] * rl[1] + s.F[i * 3 + 2] * rl[2];
          wl[i] = s.C[i * 3 + 0] * rl[0] + s.C[i * 3 + 1] * rl[1] + s.C[i * 3 + 2] * rl[2];
        }
        const double an = s.a * (rl[0] * rl[0] + rl[1] * rl[1] + rl[2] * rl[2]);
        D[0] += sc * (an + s.b * ul[0] * ul[0] + m.lam * wl[0] * wl[0]);
        D[1] += sc * (s.b * ul[0] * ul[1] + m.lam * wl[0] * wl[1]);
        D[2] += sc * (s.b * ul[0] * ul[2] + m.lam * wl[0] * wl[2]);
        D[3] += sc * (an + s.b * ul[1] * ul[1] + m.lam * wl[1] * wl[1]);
        D[4] += sc * (s.b * ul[1] * ul[2] + m.lam * wl[1] * wl[2]);
        D[5] += sc * (an + s.b * ul[2] * ul[2] + m.lam * wl[2] * wl[2]);
        const int l2 = nv < 0 ? -1 : (v[0] == nv ? 0 : (v[1] == nv ? 1 : (v[2] == nv ? 2 : (v[3] == nv ? 3 : -1))));
        if (l2 >= 0) {  // this tet also holds the chain successor: its share of the block (v, next)
          TetBlocks tb;
          tet_blocks(s, r, tb);
          double Et[9];
          tet_block_offdiag(m, s, r, tb, ul, wl, rl, l2, Et);
#pragma unroll
          for (int k = 0; k < 9; ++k) E[k] += sc * Et[k];
        }
      }
      double* q = lagw + (size_t)vtx * 16;
#pragma unroll
      for (int k = 0; k < 6; ++k) q[k] = D[k];
#pragma unroll
      for (int k = 0; k < 9; ++k) q[6 + k] = E[k];
      q[15] = 0.0;
    }
  }
}

// Launch order of the envs for the next Newton launch: counting sort (descending) by the work of the env's previous step,
// key = PCG iterations + 6 per Newton iteration (gradient, block assembly and line search cost about six sweeps), from the
// step_info rows the previous tacex_fem_step left behind (zeros before the first step: index order).  One workgroup; the order
// inside a bucket is whatever the atomics give - it only decides WHEN an env runs, never what it computes.
__global__ __launch_bounds__(1024) void fem_env_order_kernel(const double* __restrict__ step_info, int B, int* __restrict__ order) {
  constexpr int kKeys = 2048;
  __shared__ int hist[kKeys], start[kKeys];
  for (int k = threadIdx.x; k < kKeys; k += blockDim.x) hist[k] = 0;
  __syncthreads();
  auto key_of = [&](int b) {
    const double w = step_info[(size_t)b * 4 + 3] + 6.0 * step_info[(size_t)b * 4 + 0];
    return (w >= 0.0 && w < (double)(kKeys - 1)) ? (int)w : (w >= (double)(kKeys - 1) ? kKeys - 1 : 0);  // (NaN -> 0)
  };
  for (int b = threadIdx.x; b < B; b += blockDim.x) atomicAdd(&hist[key_of(b)], 1);
  __syncthreads();
  if (threadIdx.x < 64) {  // exclusive scan from the heaviest key down, one wave: 32 keys per lane + a lane scan
    const int lane = threadIdx.x;
    int loc = 0;
    for (int k = 0; k < kKeys / 64; ++k) loc += hist[kKeys - 1 - (lane * (kKeys / 64) + k)];
    int inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    int acc = inc - loc;
    for (int k = 0; k < kKeys / 64; ++k) {
      const int key = kKeys - 1 - (lane * (kKeys / 64) + k);
      start[key] = acc;
      acc += hist[key];
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) order[atomicAdd(&start[key_of(b)], 1)] = b;
}

// backward-Euler predictor of tacex_fem_step: x_prev = x, x_tilde = x + dt v + dt^2 g (US:250-252: what world.advance() starts from)
__global__ __launch_bounds__(256) void fem_predict_kernel(const double* __restrict__ x, const double* __restrict__ v, double* __restrict__ xt,
                                                          double* __restrict__ xprev, double* __restrict__ dxg, size_t n3, int B,
                                                          double dt, double g0, double g1, double g2, const double* __restrict__ ind,
                                                          const double* __restrict__ ind_prev, double* __restrict__ disp, int have_prev) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)B && dxg) dxg[i] = INFINITY;
  if (i < (size_t)B && ind && disp) {  // how far the env's indenter moved since the last step (friction slides relative to it)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      // (a NaN in the env's previous position = "none": tacex_fem_reset_envs marks a reset env so - wherever the caller puts its
      //  indenter before the next step, friction sees no sliding in that step, like the first step of a fresh scene)
      const double d = have_prev ? ind[i * 8 + 1 + k] - ind_prev[i * 3 + k] : 0.0;
      disp[i * 3 + k] = d == d ? d : 0.0;
    }
  }
  if (i >= n3) return;
  const int k = (int)(i % 3);
  const double xi = x[i];
  xprev[i] = xi;
  xt[i] = xi + dt * v[i] + dt * dt * (k == 0 ? g0 : (k == 1 ? g1 : g2));
}
__global__ __launch_bounds__(256) void fem_velocity_kernel(const double* __restrict__ x, const double* __restrict__ xprev,
                                                           double* __restrict__ v, size_t n3, double inv_dt, const double* __restrict__ ind,
                                                           double* __restrict__ ind_prev, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n3) v[i] = (x[i] - xprev[i]) * inv_dt;
  if (i < (size_t)B && ind && ind_prev) {
#pragma unroll
    for (int k = 0; k < 3; ++k) ind_prev[i * 3 + k] = ind[i * 8 + 1 + k];
  }
}

// ------------------------------------------------------------------------------------------------
// Attachment animation (UA:364-428): per step and env, aim = R(q) offset + p for every attached vertex, written straight into
// the constraint arrays the Newton kernels read (`aim_position`, `is_constrained`) - the reference computes it with
// IsaacLab's `transform_points` in float32 on the GPU, copies it to the host and hands it to libuipc's animator callback
// (UA:365-385).  One lane per (env, attachment point); float32 rotation like the reference, widened to float64 on store.
// ------------------------------------------------------------------------------------------------
// per-env reset (tacex_fem_reset_envs): one workgroup per listed env
__global__ __launch_bounds__(256) void fem_reset_envs_kernel(const int* __restrict__ ids, const double* __restrict__ pos, const double* __restrict__ rest,
                                                             double* __restrict__ x, double* __restrict__ v, double* __restrict__ step_info,
                                                             double* __restrict__ ind_prev, int V, int B) {
  const int b = ids ? ids[blockIdx.x] : (int)blockIdx.x;
  if (b < 0 || b >= B) return;
  const size_t o = (size_t)b * V * 3;
  for (int k = threadIdx.x; k < 3 * V; k += blockDim.x) {
    x[o + k] = pos ? pos[(size_t)blockIdx.x * V * 3 + k] : rest[k];
    v[o + k] = 0.0;
  }
  if (threadIdx.x < 4 && step_info) step_info[(size_t)b * 4 + threadIdx.x] = 0.0;
  if (threadIdx.x < 3 && ind_prev) ind_prev[(size_t)b * 3 + threadIdx.x] = __builtin_nan("");
}

__global__ __launch_bounds__(128) void fem_attachment_aim_kernel(const float* __restrict__ body_pos, const float* __restrict__ body_quat,
                                                                 const float* __restrict__ offsets, const int32_t* __restrict__ idx,
                                                                 double* __restrict__ aim, uint8_t* __restrict__ constrained,
                                                                 double* __restrict__ aim_compact, int A, int V) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (a >= A) return;
  const float qw = body_quat[b * 4 + 0], qx = body_quat[b * 4 + 1], qy = body_quat[b * 4 + 2], qz = body_quat[b * 4 + 3];
  // isaaclab.utils.math.matrix_from_quat: two_s = 2 / |q|^2, rows of R
  const float two_s = 2.0f / (qw * qw + qx * qx + qy * qy + qz * qz);
  const float r00 = 1.0f - two_s * (qy * qy + qz * qz), r01 = two_s * (qx * qy - qz * qw), r02 = two_s * (qx * qz + qy * qw);
  const float r10 = two_s * (qx * qy + qz * qw), r11 = 1.0f - two_s * (qx * qx + qz * qz), r12 = two_s * (qy * qz - qx * qw);
  const float r20 = two_s * (qx * qz - qy * qw), r21 = two_s * (qy * qz + qx * qw), r22 = 1.0f - two_s * (qx * qx + qy * qy);
  const float ox = offsets[a * 3 + 0], oy = offsets[a * 3 + 1], oz = offsets[a * 3 + 2];
  const float x = (r00 * ox + r01 * oy + r02 * oz) + body_pos[b * 3 + 0];
  const float y = (r10 * ox + r11 * oy + r12 * oz) + body_pos[b * 3 + 1];
  const float z = (r20 * ox + r21 * oy + r22 * oz) + body_pos[b * 3 + 2];
  const int v = idx[a];
  double* o = aim + ((size_t)b * V + v) * 3;
  o[0] = (double)x; o[1] = (double)y; o[2] = (double)z;
  constrained[(size_t)b * V + v] = 1;
  if (aim_compact) {
    double* c = aim_compact + ((size_t)b * A + a) * 3;
    c[0] = (double)x; c[1] = (double)y; c[2] = (double)z;
  }
}

// ---- K18: FEM-driven markers: barycentric surface point + pinhole projection (VT:347-366) ------------------------
__global__ __launch_bounds__(128) void fem_marker_uv_kernel(const double* __restrict__ pos, const int* __restrict__ tri,
                                                            const double* __restrict__ wgt, double fx, double fy,
                                                            double cx, double cy, double* __restrict__ uv, int Vs, int M) {
  const int mi = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (mi >= M) return;
  const double* p = pos + (size_t)b * Vs * 3;
  double q[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int v = tri[mi * 3 + k];
    const double w = wgt[mi * 3 + k];
    q[0] += w * p[v * 3 + 0]; q[1] += w * p[v * 3 + 1]; q[2] += w * p[v * 3 + 2];
  }
  uv[((size_t)b * M + mi) * 2 + 0] = fx * q[0] / q[2] + cx;
  uv[((size_t)b * M + mi) * 2 + 1] = fy * q[1] / q[2] + cy;
}

// The whole of gen_marker_flow's per-step part (VT:354-413, the static marker grid of the shipped cfgs) in ONE launch, one workgroup per env:
// surface vertices out of the FEM state -> camera frame (VT:142-187: R_inv (x - cam_pos)) -> barycentric point -> pinhole projection of ALL
// M markers (kept: `curr_marker_uv`), then the step's subset: flow[b, 0, k] = init_uv[b, sel[k]], flow[b, 1, k] = uv[b, sel[k]], optionally
// normalised (VT:407-409: / (W / 2) - 1), as float64 and / or float32 (the plugin's marker_data).  Replaces thirteen launches (index, subtract,
// batched GEMM, contiguous copy, projection, two gathers, stack, normalise, cast; 120 us of C4's 1.27 ms step: profiles/r06_experiments.md 11).
__global__ __launch_bounds__(256) void fem_marker_flow_kernel(const double* __restrict__ xg, const long long* __restrict__ surf_ids,
                                                              const double* __restrict__ cam_pos, const double* __restrict__ cam_rot_inv,
                                                              const int* __restrict__ tri, const double* __restrict__ wgt, double fx, double fy,
                                                              double cx, double cy, const double* __restrict__ init_uv,
                                                              const long long* __restrict__ sel, double norm_div, double* __restrict__ curr_uv,
                                                              double* __restrict__ flow, float* __restrict__ flow32, int V, int M, int K) {
  extern __shared__ double muv[];  // (M,2) this env's projections
  const int b = blockIdx.x;
  const double* x = xg + (size_t)b * V * 3;
  const double* cp = cam_pos + (size_t)b * 3;
  const double* R = cam_rot_inv + (size_t)b * 9;
  for (int mi = threadIdx.x; mi < M; mi += blockDim.x) {
    double q[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const long long v = surf_ids[tri[mi * 3 + k]];
      const double w = wgt[mi * 3 + k];
      const double d0 = x[v * 3] - cp[0], d1 = x[v * 3 + 1] - cp[1], d2 = x[v * 3 + 2] - cp[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) q[i] += w * (R[i * 3] * d0 + R[i * 3 + 1] * d1 + R[i * 3 + 2] * d2);
    }
    const double u = fx * q[0] / q[2] + cx, vv = fy * q[1] / q[2] + cy;
    muv[mi * 2] = u; muv[mi * 2 + 1] = vv;
    if (curr_uv) { curr_uv[((size_t)b * M + mi) * 2] = u; curr_uv[((size_t)b * M + mi) * 2 + 1] = vv; }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * K; k += blockDim.x) {
    const int which = k / K, kk = k - which * K;  // 0: initial, 1: current
    const long long s = sel[kk];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double val = which == 0 ? init_uv[((size_t)b * M + s) * 2 + c] : muv[s * 2 + c];
      if (norm_div > 0.0) val = val / norm_div - 1.0;
      const size_t o = (((size_t)b * 2 + which) * K + kk) * 2 + c;
      if (flow) flow[o] = val;
      if (flow32) flow32[o] = (float)val;
    }
  }
}

}  // namespace tacex

using namespace tacex;

struct tacex_fem_ctx {
  int device = 0;
  FemDev dev{};
  FemDev dev_nwt{};  // same mesh with the tets renumbered for fem_newton_lds_kernel (see tacex_fem_create)
  // Friction slides relative to the indenter's displacement since the previous tacex_fem_step.  The previous positions (B,3) live
  // in the caller's workspace (no allocation in a compute call); the context remembers WHICH workspace and env count they were
  // written for, so another workspace or num_envs never reads stale data, and a new indenter TENSOR for the same contact (the
  // caller animating the indenter by passing fresh tensors) keeps them: only disabling contact or its first enable resets.
  const void* ind_prev_ws = nullptr;
  int ind_prev_B = 0;
  const double* rest = nullptr;  // (V,3) rest positions (tacex_fem_reset_envs)
  int last_resident = -1;       // which Newton kernel the last launch used: 1 CU-resident, 0 streaming, -1 none yet (tacex_fem_newton_resident)
  int ls_refine = 4;            // bisections after a cut line search in fem_ball_newton_kernel (tacex_fem_set_line_search_refine)
  int fric_lag_mode = 0;        // 0: lag at the step's start state, capped by the contact reaction (round 4); 1: IPC's previous-configuration lag
  bool deterministic = false;   // window + CSR-gather sweeps (fixed summation order) instead of LDS atomics (tacex_fem_set_deterministic)
  bool follow_indenter = true;  // contact-following start of the Newton loop (tacex_fem_set_contact_following; fem_newton_lds_kernel)
  double* dx_dev = nullptr;  // optional (B,) last Newton update max|dx| per env: converged envs skip further iterations
  double dx_tol = 0.0;
  tacex::BallDev ball{};     // the env's free affine body + ground (tacex_fem_set_affine_body); nv == 0: none
  const void* ball_last_ws = nullptr;  // workspace / env count whose "q at the end of the previous step" rows are valid (kinematic bodies)
  int ball_last_B = 0;
  std::vector<void*> allocs;
};

// Tables a setter REPLACES (coarse space, chains, indenter mesh, vertex areas) are freed once nothing can read them any more:
// the setters are host-synchronous set-up calls, so they drain the device first (a caller who refreshed the
// preconditioner every step grew the device memory without bound).
template <typename T>
static void fem_release(tacex_fem_ctx* c, const T*& ptr, bool* synced) {
  if (!ptr) return;
  if (!*synced) { (void)hipDeviceSynchronize(); *synced = true; }
  void* p = const_cast<void*>(static_cast<const void*>(ptr));
  for (size_t i = 0; i < c->allocs.size(); ++i)
    if (c->allocs[i] == p) { c->allocs.erase(c->allocs.begin() + (long)i); (void)hipFree(p); break; }
  ptr = nullptr;
}

template <typename T>
static int fem_upload(tacex_fem_ctx* c, const std::vector<T>& h, const T** out) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, h.size() * sizeof(T) + 16);
  if (e != hipSuccess) return fail_hip(e, "hipMalloc(fem table)");
  c->allocs.push_back(p);
  e = hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail_hip(e, "hipMemcpy(fem table)");
  *out = static_cast<const T*>(p);
  return 0;
}

// signed distance of every vertex to its env's indenter (the solver's own distance function): what a caller needs to keep the
// "approach by less than the gap" contract of tacex_fem_step without restating the indenter geometry
__global__ __launch_bounds__(256) void fem_contact_gaps_kernel(FemDev m, const double* xg, double* gaps) {
  const int b = blockIdx.x;
  const double* ind = m.indenters ? m.indenters + (size_t)b * 8 : nullptr;
  for (int v = threadIdx.x; v < m.V; v += blockDim.x) {
    const double* x = xg + ((size_t)b * m.V + v) * 3;
    const double x3[3] = {x[0], x[1], x[2]};
    const ContactEval c = contact_eval<true>(m, ind, 1.0, x3);
    gaps[(size_t)b * m.V + v] = c.d < 1e299 ? c.d : INFINITY;
  }
}

// unique edges (lower index first) of a triangle list, the area each stands for (a third of its triangles' rest areas: the edge areas sum
// to the surface area) and its squared rest length - oracle/abd_oracle.py surface_edges
static void surface_edges(const std::vector<int>& tri, const double* X, std::vector<int>& edges, std::vector<double>& earea, std::vector<double>& elen2) {
  std::map<std::pair<int, int>, double> acc;
  for (size_t t = 0; t + 2 < tri.size(); t += 3) {
    const double* a = X + (size_t)tri[t] * 3; const double* b = X + (size_t)tri[t + 1] * 3; const double* cc = X + (size_t)tri[t + 2] * 3;
    const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]};
    const double cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double ta = 0.5 * sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    for (int k = 0; k < 3; ++k) {
      const int u = tri[t + k], v = tri[t + (k + 1) % 3];
      acc[{u < v ? u : v, u < v ? v : u}] += ta / 3.0;
    }
  }
  edges.clear(); earea.clear(); elen2.clear();
  for (const auto& kv : acc) {
    const int u = kv.first.first, v = kv.first.second;
    edges.push_back(u); edges.push_back(v);
    earea.push_back(kv.second);
    double l2 = 0.0;
    for (int i = 0; i < 3; ++i) { const double dd = X[(size_t)v * 3 + i] - X[(size_t)u * 3 + i]; l2 += dd * dd; }
    elen2.push_back(l2);
  }
}

extern "C" {

int tacex_fem_create(int device_id, const tacex_fem_params* p, tacex_fem_ctx** out) {
  if (!p || !out || !p->rest_positions || !p->tets) { set_error("tacex_fem_create: null argument"); return 2; }
  const int V = p->num_verts, T = p->num_tets;
  if (V < 4 || T < 1) { set_error("tacex_fem_create: mesh too small (V=%d, T=%d)", V, T); return 2; }
  if (!(p->poisson > -1.0 && p->poisson < 0.5) || !(p->youngs > 0.0) || !(p->density > 0.0) || !(p->dt > 0.0)) {
    set_error("tacex_fem_create: need youngs > 0, -1 < poisson < 0.5, density > 0, dt > 0");
    return 2;
  }
  std::vector<int> tets((size_t)4 * T);
  std::vector<double> dminv((size_t)9 * T), vol(T), mass(V, 0.0);
  const double* X = p->rest_positions;
  for (int t = 0; t < T; ++t) {
    int v[4];
    for (int k = 0; k < 4; ++k) {
      v[k] = p->tets[t * 4 + k];
      if (v[k] < 0 || v[k] >= V) { set_error("tacex_fem_create: tet %d has vertex index %d out of range", t, v[k]); return 2; }
    }
    double Dm[9], det = 0.0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) Dm[i * 3 + k] = X[v[k + 1] * 3 + i] - X[v[0] * 3 + i];
      det = Dm[0] * (Dm[4] * Dm[8] - Dm[5] * Dm[7]) - Dm[1] * (Dm[3] * Dm[8] - Dm[5] * Dm[6]) + Dm[2] * (Dm[3] * Dm[7] - Dm[4] * Dm[6]);
      if (det > 0.0) break;
      int tmp = v[1]; v[1] = v[2]; v[2] = tmp;  // re-orient (same rule as the oracle)
    }
    if (!(det > 0.0)) { set_error("tacex_fem_create: tet %d is degenerate", t); return 2; }
    const double id = 1.0 / det;
    double inv[9] = {(Dm[4] * Dm[8] - Dm[5] * Dm[7]) * id, (Dm[2] * Dm[7] - Dm[1] * Dm[8]) * id, (Dm[1] * Dm[5] - Dm[2] * Dm[4]) * id,
                     (Dm[5] * Dm[6] - Dm[3] * Dm[8]) * id, (Dm[0] * Dm[8] - Dm[2] * Dm[6]) * id, (Dm[2] * Dm[3] - Dm[0] * Dm[5]) * id,
                     (Dm[3] * Dm[7] - Dm[4] * Dm[6]) * id, (Dm[1] * Dm[6] - Dm[0] * Dm[7]) * id, (Dm[0] * Dm[4] - Dm[1] * Dm[3]) * id};
    for (int k = 0; k < 9; ++k) dminv[(size_t)k * T + t] = inv[k];
    for (int k = 0; k < 4; ++k) tets[(size_t)k * T + t] = v[k];
    vol[t] = det / 6.0;
    for (int k = 0; k < 4; ++k) mass[v[k]] += p->density * vol[t] / 4.0;
  }
  std::vector<int> off(V + 1, 0), idx((size_t)4 * T);
  for (int t = 0; t < T; ++t)
    for (int k = 0; k < 4; ++k) off[tets[(size_t)k * T + t] + 1]++;
  for (int v = 0; v < V; ++v) off[v + 1] += off[v];
  {
    std::vector<int> cur(off.begin(), off.end() - 1);
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < 4; ++k) idx[cur[tets[(size_t)k * T + t]]++] = t * 4 + k;
  }
  hipError_t e = hipSetDevice(device_id);
  if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
  auto* c = new tacex_fem_ctx();
  c->device = device_id;
  FemDev& d = c->dev;
  d.V = V; d.T = T;
  {
    std::vector<double> rest(X, X + (size_t)3 * V);
    if (int rc0 = fem_upload(c, rest, &c->rest)) { tacex_fem_destroy(c); return rc0; }
  }
  int rc = fem_upload(c, tets, &d.tets) | fem_upload(c, dminv, &d.dminv) | fem_upload(c, vol, &d.vol) |
           fem_upload(c, mass, &d.mass) | fem_upload(c, off, &d.vt_off) | fem_upload(c, idx, &d.vt_idx);
  // Renumbered copy for the LDS Newton kernel.  Its per-window vertex gather waits, in every window, for the vertex with
  // the most incident tets IN THAT WINDOW; with a spatially coherent numbering all ~24 tets of a vertex sit in one or two
  // windows, so each of the 4 windows costs a full-valence gather (measured: 54 % of the kernel).  Dealing the tets out
  // round-robin spreads every vertex's tets evenly over the windows.
  {
    const int nwin = (T + kNwtChunk - 1) / kNwtChunk;
    std::vector<int> order;
    order.reserve(T);
    for (int w = 0; w < nwin; ++w)
      for (int t = w; t < T; t += nwin) order.push_back(t);
    std::vector<int> tets2((size_t)4 * T), off2(off), idx2((size_t)4 * T);
    std::vector<double> dminv2((size_t)9 * T), vol2(T);
    for (int n = 0; n < T; ++n) {
      const int t = order[n];
      for (int k = 0; k < 4; ++k) tets2[(size_t)k * T + n] = tets[(size_t)k * T + t];
      for (int k = 0; k < 9; ++k) dminv2[(size_t)k * T + n] = dminv[(size_t)k * T + t];
      vol2[n] = vol[t];
    }
    std::vector<int> cur(off.begin(), off.end() - 1);
    for (int n = 0; n < T; ++n)
      for (int k = 0; k < 4; ++k) idx2[cur[tets2[(size_t)k * T + n]]++] = n * 4 + k;
    std::vector<double> rec2((size_t)12 * T);
    for (int n = 0; n < T; ++n) {
      int ids[4];
      for (int k = 0; k < 4; ++k) ids[k] = tets2[(size_t)k * T + n];
      memcpy(&rec2[(size_t)n * 12], ids, sizeof(ids));  // 4 ints = the first two doubles of the record
      for (int k = 0; k < 9; ++k) rec2[(size_t)n * 12 + 2 + k] = dminv2[(size_t)k * T + n];
      rec2[(size_t)n * 12 + 11] = vol2[n];
    }
    const int nblk = (T + 63) / 64;
    std::vector<double> blk2((size_t)nblk * (kTetBlkBytes / 8), 0.0);
    for (int n = 0; n < T; ++n) {
      char* base = reinterpret_cast<char*>(blk2.data()) + (size_t)(n >> 6) * kTetBlkBytes;
      const int ln = n & 63;
      for (int k = 0; k < 9; ++k) memcpy(base + k * 512 + ln * 8, &dminv2[(size_t)k * T + n], 8);
      memcpy(base + 4608 + ln * 8, &vol2[n], 8);
      for (int k = 0; k < 4; ++k) memcpy(base + 5120 + k * 256 + ln * 4, &tets2[(size_t)k * T + n], 4);
    }
    c->dev_nwt = d;
    rc = fem_upload(c, blk2, &c->dev_nwt.tet_blk) | fem_upload(c, tets2, &c->dev_nwt.tets) | fem_upload(c, dminv2, &c->dev_nwt.dminv) | fem_upload(c, vol2, &c->dev_nwt.vol) |
         fem_upload(c, idx2, &c->dev_nwt.vt_idx) | fem_upload(c, rec2, &c->dev_nwt.tet_rec);
  }
  if (rc) { tacex_fem_destroy(c); return rc; }
  const double mu_l = p->youngs / (2.0 * (1.0 + p->poisson));
  const double lam_l = p->youngs * p->poisson / ((1.0 + p->poisson) * (1.0 - 2.0 * p->poisson));
  d.mu = 4.0 / 3.0 * mu_l;
  d.lam = lam_l + 5.0 / 6.0 * mu_l;
  d.alpha = 1.0 + 0.75 * d.mu / d.lam;
  d.psi_rest = 0.5 * d.lam * (1.0 - d.alpha) * (1.0 - d.alpha) - 0.5 * d.mu * log(4.0);
  d.dt = p->dt;
  d.strength = p->constraint_strength_ratio;
  {
    double lo[3] = {X[0], X[1], X[2]}, hi[3] = {X[0], X[1], X[2]};
    for (int v = 1; v < V; ++v)
      for (int i = 0; i < 3; ++i) { lo[i] = fmin(lo[i], X[v * 3 + i]); hi[i] = fmax(hi[i], X[v * 3 + i]); }
    d.step_cap = sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    c->dev_nwt.step_cap = d.step_cap;
  }
  c->dev_nwt.mu = d.mu; c->dev_nwt.lam = d.lam; c->dev_nwt.alpha = d.alpha; c->dev_nwt.psi_rest = d.psi_rest;
  c->dev_nwt.dt = d.dt; c->dev_nwt.strength = d.strength;
  *out = c;
  return 0;
}

void tacex_fem_destroy(tacex_fem_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (void* p : c->allocs) (void)hipFree(p);
  delete c;
}

size_t tacex_fem_workspace_bytes(const tacex_fem_ctx* c, int B) {
  if (!c || B <= 0) return 0;
  // env blocks | x_prev (B,V,3) | max |d| (B) | indenter displacement (B,3) | previous indenter position (B,3)
  // ... | env launch order (B int32, rounded up to doubles)
  return ((size_t)B * newton_ws_doubles(c->dev.V, c->dev.T) + (size_t)B * 3 * c->dev.V + (size_t)7 * B + 8 + ((size_t)B + 1) / 2) * sizeof(double);
}

int tacex_fem_element_terms(tacex_fem_ctx* c, const double* x, double* energy, double* grad, double* hess,
                            int project_psd, int B, void* stream) {
  if (!c || !x) { set_error("tacex_fem_element_terms: null argument"); return 2; }
  if (B <= 0) return 0;
  if (project_psd && hess)
    hipLaunchKernelGGL(fem_element_terms_kernel<true>, dim3((c->dev.T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       c->dev, x, energy, grad, hess);
  else
    hipLaunchKernelGGL(fem_element_terms_kernel<false>, dim3((c->dev.T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       c->dev, x, energy, grad, hess);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_element_terms_kernel");
}

int tacex_fem_energy(tacex_fem_ctx* c, const double* x, const double* xt, const uint8_t* cons, const double* aim,
                     double* E, void* ws, int B, void* stream) {
  (void)ws;
  if (!c || !x || !xt || !E) { set_error("tacex_fem_energy: null argument"); return 2; }
  if ((cons == nullptr) != (aim == nullptr)) { set_error("tacex_fem_energy: constrained_dev and aim_dev go together"); return 2; }
  if (B <= 0) return 0;
  hipLaunchKernelGGL(fem_energy_kernel, dim3(B), dim3(512), 0, (hipStream_t)stream, c->dev, x, xt, cons, aim, E);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_energy_kernel");
}

int tacex_fem_gradient(tacex_fem_ctx* c, const double* x, const double* xt, const uint8_t* cons, const double* aim,
                       double* g, void* ws, int B, void* stream) {
  if (!c || !x || !xt || !g || !ws) { set_error("tacex_fem_gradient: null argument"); return 2; }
  if ((cons == nullptr) != (aim == nullptr)) { set_error("tacex_fem_gradient: constrained_dev and aim_dev go together"); return 2; }
  if (B <= 0) return 0;
  hipLaunchKernelGGL(fem_gradient_kernel, dim3(B), dim3(512), 0, (hipStream_t)stream, c->dev, x, xt, cons, aim, g,
                     static_cast<double*>(ws));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_gradient_kernel");
}

int tacex_fem_set_contact(tacex_fem_ctx* c, const double* vertex_area_host, double d_hat, double stiffness, const double* indenters_dev) {
  if (!c) { set_error("tacex_fem_set_contact: null context"); return 2; }
  if (vertex_area_host) {
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    std::vector<double> a(vertex_area_host, vertex_area_host + c->dev.V);
    for (double v : a)
      if (!(v >= 0.0)) { set_error("tacex_fem_set_contact: vertex areas must be >= 0"); return 2; }
    const double* d = nullptr;
    if (int rc = fem_upload(c, a, &d)) return rc;
    bool synced = false;
    fem_release(c, c->dev.area, &synced);
    c->dev.area = d; c->dev_nwt.area = d;
  }
  if (indenters_dev) {
    if (!c->dev.area) { set_error("tacex_fem_set_contact: vertex areas were never given"); return 2; }
    if (!(d_hat > 0.0) || !(stiffness > 0.0)) { set_error("tacex_fem_set_contact: need d_hat > 0 and stiffness > 0"); return 2; }
  }
  c->dev.dhat = c->dev_nwt.dhat = d_hat;
  c->dev.kappa = c->dev_nwt.kappa = stiffness;
  if (!indenters_dev || !c->dev.indenters) { c->ind_prev_ws = nullptr; c->ind_prev_B = 0; }  // contact off, or enabled for the first time:
                                                                                             // the next step sees no indenter motion
  c->dev.indenters = c->dev_nwt.indenters = indenters_dev;
  return 0;
}

int tacex_fem_contact_gaps(tacex_fem_ctx* c, const double* x_dev, double* gaps_dev, int num_envs, void* stream) {
  if (!c || !x_dev || !gaps_dev) { set_error("tacex_fem_contact_gaps: null argument"); return 2; }
  if (num_envs <= 0) return 0;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
  hipLaunchKernelGGL(fem_contact_gaps_kernel, dim3(num_envs), dim3(256), 0, (hipStream_t)stream, c->dev, x_dev, gaps_dev);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_contact_gaps_kernel");
}

int tacex_fem_set_deterministic(tacex_fem_ctx* c, int enable) {
  if (!c) { set_error("tacex_fem_set_deterministic: null context"); return 2; }
  c->deterministic = enable != 0;
  return 0;
}

int tacex_fem_set_contact_following(tacex_fem_ctx* c, int enable) {
  if (!c) { set_error("tacex_fem_set_contact_following: null context"); return 2; }
  c->follow_indenter = enable != 0;
  return 0;
}

int tacex_fem_set_friction(tacex_fem_ctx* c, double friction_ratio, double eps_velocity) {
  if (!c) { set_error("tacex_fem_set_friction: null context"); return 2; }
  if (!(friction_ratio >= 0.0) || (friction_ratio > 0.0 && !(eps_velocity > 0.0))) {
    set_error("tacex_fem_set_friction: need friction_ratio >= 0 and eps_velocity > 0");
    return 2;
  }
  c->dev.fric_mu = c->dev_nwt.fric_mu = friction_ratio;
  c->dev.fric_eps = c->dev_nwt.fric_eps = eps_velocity * c->dev.dt;
  return 0;
}

int tacex_fem_set_line_search_refine(tacex_fem_ctx* c, int bisections) {
  if (!c || bisections < 0 || bisections > 15) { set_error("tacex_fem_set_line_search_refine: need a context and 0 <= bisections <= 15"); return 2; }
  c->ls_refine = bisections;
  return 0;
}

int tacex_fem_set_friction_lag(tacex_fem_ctx* c, int mode) {
  if (!c) { set_error("tacex_fem_set_friction_lag: null context"); return 2; }
  if (mode != 0 && mode != 1) { set_error("tacex_fem_set_friction_lag: mode must be 0 (reaction-capped lag at the step's start) or 1 (IPC: previous configuration)"); return 2; }
  c->fric_lag_mode = mode;
  return 0;
}

int tacex_fem_newton_resident(const tacex_fem_ctx* c) { return c ? c->last_resident : -1; }

int tacex_fem_set_indenter_mesh(tacex_fem_ctx* c, int num_verts, const double* verts_host, int num_tris, const int32_t* tris_host) {
  if (!c) { set_error("tacex_fem_set_indenter_mesh: null context"); return 2; }
  if (num_tris == 0) {
    bool synced = false;
    (void)hipSetDevice(c->device);
    fem_release(c, c->dev.im_tri, &synced); fem_release(c, c->dev.im_bs, &synced); fem_release(c, c->dev.im_cl, &synced);
    c->dev_nwt.im_tri = nullptr; c->dev_nwt.im_bs = nullptr; c->dev_nwt.im_cl = nullptr;
    c->dev.im_nt = c->dev_nwt.im_nt = 0;
    return 0;
  }
  if (num_verts < 3 || num_tris < 1 || !verts_host || !tris_host) { set_error("tacex_fem_set_indenter_mesh: bad arguments"); return 2; }
  for (int k = 0; k < num_tris * 3; ++k)
    if (tris_host[k] < 0 || tris_host[k] >= num_verts) { set_error("tacex_fem_set_indenter_mesh: vertex index %d out of range", tris_host[k]); return 2; }
  // triangles in Morton order of their centroids (10 bits per axis over the bounding box): 16 consecutive ones form a cluster
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < num_verts; ++i)
    for (int k = 0; k < 3; ++k) { lo[k] = fmin(lo[k], verts_host[(size_t)i * 3 + k]); hi[k] = fmax(hi[k], verts_host[(size_t)i * 3 + k]); }
  std::vector<std::pair<uint32_t, int>> order(num_tris);
  for (int t = 0; t < num_tris; ++t) {
    uint32_t code = 0;
    for (int k = 0; k < 3; ++k) {
      double cen = 0.0;
      for (int j = 0; j < 3; ++j) cen += verts_host[(size_t)tris_host[(size_t)t * 3 + j] * 3 + k] / 3.0;
      const double ext = hi[k] - lo[k];
      uint32_t q = ext > 0.0 ? (uint32_t)fmin(1023.0, fmax(0.0, (cen - lo[k]) / ext * 1024.0)) : 0u;
      for (int bit = 0; bit < 10; ++bit) code |= ((q >> bit) & 1u) << (3 * bit + k);
    }
    order[t] = {code, t};
  }
  std::sort(order.begin(), order.end());
  const int ncl = (num_tris + kMeshCluster - 1) / kMeshCluster;
  std::vector<double> tri((size_t)num_tris * 9), bs((size_t)num_tris * 4), cl((size_t)ncl * 4);
  for (int s = 0; s < num_tris; ++s) {
    const int t = order[s].second;
    const double* v[3];
    for (int k = 0; k < 3; ++k) v[k] = verts_host + (size_t)tris_host[(size_t)t * 3 + k] * 3;
    double cen[3];
    for (int k = 0; k < 3; ++k) {
      tri[(size_t)s * 9 + k] = v[0][k];
      tri[(size_t)s * 9 + 3 + k] = v[1][k] - v[0][k];
      tri[(size_t)s * 9 + 6 + k] = v[2][k] - v[0][k];
      cen[k] = (v[0][k] + v[1][k] + v[2][k]) / 3.0;
      bs[(size_t)s * 4 + k] = cen[k];
    }
    double r = 0.0;
    for (int j = 0; j < 3; ++j) {
      const double d0 = v[j][0] - cen[0], d1 = v[j][1] - cen[1], d2 = v[j][2] - cen[2];
      r = fmax(r, sqrt(d0 * d0 + d1 * d1 + d2 * d2));
    }
    bs[(size_t)s * 4 + 3] = r * (1.0 + 1e-12);
  }
  for (int q = 0; q < ncl; ++q) {
    const int s0 = q * kMeshCluster, s1 = std::min(num_tris, s0 + kMeshCluster);
    double cen[3] = {0, 0, 0};
    for (int s = s0; s < s1; ++s)
      for (int k = 0; k < 3; ++k) cen[k] += bs[(size_t)s * 4 + k] / (s1 - s0);
    double r = 0.0;
    for (int s = s0; s < s1; ++s) {
      const double d0 = bs[(size_t)s * 4] - cen[0], d1 = bs[(size_t)s * 4 + 1] - cen[1], d2 = bs[(size_t)s * 4 + 2] - cen[2];
      r = fmax(r, sqrt(d0 * d0 + d1 * d1 + d2 * d2) + bs[(size_t)s * 4 + 3]);
    }
    for (int k = 0; k < 3; ++k) cl[(size_t)q * 4 + k] = cen[k];
    cl[(size_t)q * 4 + 3] = r * (1.0 + 1e-12);
  }
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
  FemDev& d = c->dev;
  {
    bool synced = false;
    fem_release(c, d.im_tri, &synced); fem_release(c, d.im_bs, &synced); fem_release(c, d.im_cl, &synced);
    d.im_nt = c->dev_nwt.im_nt = 0;
  }
  if (int rc = fem_upload(c, tri, &d.im_tri) | fem_upload(c, bs, &d.im_bs) | fem_upload(c, cl, &d.im_cl)) return rc;
  d.im_nt = num_tris;
  c->dev_nwt.im_nt = num_tris; c->dev_nwt.im_tri = d.im_tri; c->dev_nwt.im_bs = d.im_bs; c->dev_nwt.im_cl = d.im_cl;
  return 0;
}

int tacex_fem_set_chains(tacex_fem_ctx* c, int num_chains, const int32_t* chain_offsets_host, const int32_t* chain_vertices_host) {
  if (!c) { set_error("tacex_fem_set_chains: null context"); return 2; }
  FemDev& d = c->dev;
  FemDev& n2 = c->dev_nwt;
  auto drop_chains = [&]() {
    bool synced = false;
    (void)hipSetDevice(c->device);
    fem_release(c, d.ch_head, &synced); fem_release(c, d.ch_next, &synced); fem_release(c, d.ch_prev, &synced);
    d.nch = n2.nch = 0;
    n2.ch_head = nullptr; n2.ch_next = nullptr; n2.ch_prev = nullptr;
  };
  if (num_chains == 0) {  // every vertex its own chain: 3x3 block Jacobi
    drop_chains();
    return 0;
  }
  if (num_chains < 0 || !chain_offsets_host || !chain_vertices_host) { set_error("tacex_fem_set_chains: bad arguments"); return 2; }
  const int V = d.V;
  std::vector<int> nxt(V, -1), prv(V, -1), head;
  std::vector<char> member(V, 0);
  for (int ci = 0; ci < num_chains; ++ci) {
    const int a = chain_offsets_host[ci], b = chain_offsets_host[ci + 1];
    if (a < 0 || b < a) { set_error("tacex_fem_set_chains: offsets must ascend"); return 2; }
    for (int k = a; k < b; ++k) {
      const int v = chain_vertices_host[k];
      if (v < 0 || v >= V || member[v]) { set_error("tacex_fem_set_chains: vertex %d out of range or in two chains", v); return 2; }
      member[v] = 1;
      if (k > a) { nxt[chain_vertices_host[k - 1]] = v; prv[v] = chain_vertices_host[k - 1]; }
    }
    if (b > a) head.push_back(chain_vertices_host[a]);
  }
  for (int v = 0; v < V; ++v)
    if (!member[v]) head.push_back(v);  // the rest are chains of one vertex
  std::sort(head.begin(), head.end());
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
  drop_chains();
  if (int rc = fem_upload(c, head, &d.ch_head) | fem_upload(c, nxt, &d.ch_next) | fem_upload(c, prv, &d.ch_prev)) return rc;
  d.nch = (int)head.size();
  n2.nch = d.nch; n2.ch_head = d.ch_head; n2.ch_next = d.ch_next; n2.ch_prev = d.ch_prev;
  return 0;
}

int tacex_fem_set_coarse_space(tacex_fem_ctx* c, int num_coarse, const int32_t* vertex_nodes_host, const double* vertex_weights_host,
                               const double* coarse_inverse_host) {
  if (!c) { set_error("tacex_fem_set_coarse_space: null context"); return 2; }
  auto drop_coarse = [&]() {
    bool synced = false;
    (void)hipSetDevice(c->device);
    FemDev& d0 = c->dev;
    fem_release(c, d0.cv_node, &synced); fem_release(c, d0.cv_w, &synced); fem_release(c, d0.cn_off, &synced);
    fem_release(c, d0.cn_vtx, &synced); fem_release(c, d0.cn_w, &synced); fem_release(c, d0.ac_inv, &synced);
    FemDev& n0 = c->dev_nwt;
    n0.cv_node = nullptr; n0.cv_w = nullptr; n0.cn_off = nullptr; n0.cn_vtx = nullptr; n0.cn_w = nullptr; n0.ac_inv = nullptr;
    d0.nc = n0.nc = 0;
  };
  if (num_coarse == 0) {  // block Jacobi alone
    drop_coarse();
    return 0;
  }
  if (num_coarse < 1 || num_coarse > kFemMaxCoarse || !vertex_nodes_host || !vertex_weights_host || !coarse_inverse_host) {
    set_error("tacex_fem_set_coarse_space: need 1 <= num_coarse <= %d and non-null tables", kFemMaxCoarse);
    return 2;
  }
  const int V = c->dev.V, nc = num_coarse;
  std::vector<int> node(vertex_nodes_host, vertex_nodes_host + (size_t)V * 8);
  std::vector<double> w(vertex_weights_host, vertex_weights_host + (size_t)V * 8);
  for (size_t i = 0; i < node.size(); ++i)
    if (node[i] < 0 || node[i] >= nc || !(w[i] >= 0.0)) { set_error("tacex_fem_set_coarse_space: bad (node, weight) entry %zu", i); return 2; }
  // coarse node -> support (vertex, weight), zero weights dropped, vertices ascending (fixed summation order)
  std::vector<int> off(nc + 1, 0), vtx;
  std::vector<double> cw;
  for (int n = 0; n < nc; ++n) {
    for (int v = 0; v < V; ++v) {
      double ws = 0.0;
      for (int k = 0; k < 8; ++k)
        if (node[(size_t)v * 8 + k] == n) ws += w[(size_t)v * 8 + k];
      if (ws > 0.0) { vtx.push_back(v); cw.push_back(ws); }
    }
    off[n + 1] = (int)vtx.size();
  }
  std::vector<double> ai(coarse_inverse_host, coarse_inverse_host + (size_t)9 * nc * nc);
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
  FemDev& d = c->dev;
  drop_coarse();
  if (int rc = fem_upload(c, node, &d.cv_node) | fem_upload(c, w, &d.cv_w) | fem_upload(c, off, &d.cn_off) | fem_upload(c, vtx, &d.cn_vtx) |
               fem_upload(c, cw, &d.cn_w) | fem_upload(c, ai, &d.ac_inv))
    return rc;
  d.nc = nc;
  FemDev& n2 = c->dev_nwt;
  n2.nc = nc; n2.cv_node = d.cv_node; n2.cv_w = d.cv_w; n2.cn_off = d.cn_off; n2.cn_vtx = d.cn_vtx; n2.cn_w = d.cn_w; n2.ac_inv = d.ac_inv;
  return 0;
}

int tacex_fem_set_newton_early_exit(tacex_fem_ctx* c, double* dx_dev, double dx_tol) {
  if (!c) { set_error("tacex_fem_set_newton_early_exit: null context"); return 2; }
  c->dx_dev = dx_dev;
  c->dx_tol = dx_dev ? dx_tol : 0.0;
  return 0;
}

// one launch of the Newton kernel: up to max_newton iterations per env inside the CU-resident kernel, one iteration of the
// streaming fallback (mesh with more vertices than a workgroup has threads; TACEX_FEM_NEWTON_LDS=0)
// elastic preconditioner blocks of all envs at x -> the env blocks of the workspace ((V,16) per env); dx_dev / dx_tol: envs already converged are skipped
static int launch_assemble(tacex_fem_ctx* c, const double* x, void* ws, int B, const double* dx_dev, double dx_tol, hipStream_t st, bool atom) {
  static size_t granted_a[2][64] = {};
  // TACEX_FEM_ASSEMBLE: 1 = tet-centric with LDS atomics, 0 = vertex-centric in a fixed order (A/B hook; default: follows the atomics switch)
  static const int asm_env = getenv("TACEX_FEM_ASSEMBLE") ? atoi(getenv("TACEX_FEM_ASSEMBLE")) : -1;
  const bool asm_atom = asm_env < 0 ? atom : (asm_env != 0 && !c->deterministic);
  const int V = c->dev.V;
  const size_t lds_a = ((size_t)3 * V + (asm_atom ? (size_t)15 * V : 0)) * sizeof(double);
  auto ka = asm_atom ? fem_assemble_blocks_kernel<true> : fem_assemble_blocks_kernel<false>;
  hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(ka), lds_a, granted_a[asm_atom ? 1 : 0]);
  if (ea != hipSuccess) return fail_hip(ea, "hipFuncSetAttribute(fem_assemble_blocks_kernel)");
  hipLaunchKernelGGL(ka, dim3(B), dim3(512), lds_a, st, c->dev_nwt, x, static_cast<double*>(ws), dx_dev, dx_tol);
  ea = hipGetLastError();
  return ea == hipSuccess ? 0 : fail_hip(ea, "fem_assemble_blocks_kernel");
}

static int launch_newton(tacex_fem_ctx* c, double* x, const double* xt, const uint8_t* cons, const double* aim, double* stats, void* ws,
                         int B, int pcg_max_iter, double pcg_tol_rate, int ls_max_iter, double* dx_dev, double dx_tol, int max_newton,
                         double* step_info, hipStream_t st, bool* resident, const double* xprev = nullptr, const double* disp = nullptr,
                         const int* env_order = nullptr, bool stream_accumulate = false) {
  static const int use_lds = getenv("TACEX_FEM_NEWTON_LDS") ? atoi(getenv("TACEX_FEM_NEWTON_LDS")) : 1;
  const bool fric = xprev && disp && c->dev.indenters && c->dev.fric_mu > 0.0;
  const bool mesh = c->dev.indenters && c->dev.im_nt > 0;  // the mesh-capable instantiation only when a mesh indenter exists
  static const int env_atomic = getenv("TACEX_FEM_ATOMIC") ? atoi(getenv("TACEX_FEM_ATOMIC")) : 1;  // A/B hook
  const bool atom = env_atomic != 0 && !c->deterministic;
  static const int fric_lag_at_start = getenv("TACEX_FEM_FRIC_LAG") ? atoi(getenv("TACEX_FEM_FRIC_LAG")) : 1;  // A/B hook: 0 = lag where this step's normal contact converged
  // threads per env: one per vertex, in steps of four waves (nwt_window_doubles); the wide variants are atomic-only and take analytic
  // indenters only (a mesh indenter or the deterministic switch on a mesh of more than 512 vertices: streaming kernel below)
  const int V = c->dev.V;
  // Meshes of <= 256 vertices run on 256 threads (round 6): ONE wave per SIMD with the whole 512-register file per lane (256 VGPRs + 74 AGPRs,
  // ScratchSize 0) against two waves per SIMD and 356 B/lane of scratch on 512 threads - measured on a 210-vertex / 720-tet pad, same
  // box, alternating: 0.248 / 0.249 against 0.294 / 0.290 ms per step of 512 envs (-15 %), 0.145 / 0.151 against 0.178 / 0.174 at 256
  // envs (profiles/r06_experiments.md section 3).  TACEX_FEM_NT256=0 keeps 512 threads (A/B).  The C4 pad (495 vertices) cannot run
  // it: a thread OWNS a vertex.
  static const int nt256 = getenv("TACEX_FEM_NT256") ? atoi(getenv("TACEX_FEM_NT256")) : 1;
  const int nt = (nt256 && V <= 256 && atom && !mesh) ? 256 : (V <= 512 ? 512 : 768);
  const size_t lds = nwt_lds_bytes(V, c->dev.T, fric, nt);
  if (use_lds && V <= 768 && (nt <= 512 ? 4 * c->dev.T < 65535 : (atom && !mesh)) && lds <= 160 * 1024) {
    if (resident) *resident = true;
    c->last_resident = 1;
    static size_t granted[6][64] = {};  // per kernel instantiation and device: the attribute is per kernel AND device
    using kern_t = decltype(&fem_newton_lds_kernel<false, true, 512>);
    kern_t kern;
    int slot;
    if (nt == 256) {
      kern = fem_newton_lds_kernel<false, true, 256>;
      slot = 5;
    } else if (nt == 512) {
      kern = mesh ? (atom ? fem_newton_lds_kernel<true, true, 512> : fem_newton_lds_kernel<true, false, 512>)
                  : (atom ? fem_newton_lds_kernel<false, true, 512> : fem_newton_lds_kernel<false, false, 512>);
      slot = (mesh ? 1 : 0) + (atom ? 2 : 0);
    } else {
      kern = fem_newton_lds_kernel<false, true, 768>;
      slot = 4;
    }
    hipError_t ea = hipSetDevice(c->device);
    if (ea == hipSuccess) ea = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, granted[slot]);
    if (ea != hipSuccess) return fail_hip(ea, "hipFuncSetAttribute(fem_newton_lds_kernel)");
    // elastic preconditioner blocks of all envs at the state this launch starts from -> workspace ((V,16) per env)
    if (int rc = launch_assemble(c, x, ws, B, dx_dev, dx_tol, st, atom)) return rc;
    hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, st, c->dev_nwt, x, xt, cons, aim, stats, pcg_max_iter,
                       pcg_tol_rate, ls_max_iter, dx_dev, dx_tol, max_newton, step_info, fric ? xprev : nullptr,
                       (xprev && disp && c->dev.indenters) ? disp : nullptr, env_order,
                       (c->follow_indenter ? 1 : 0) | ((fric_lag_at_start || c->fric_lag_mode == 1) ? 2 : 0) | (c->fric_lag_mode == 1 ? 4 : 0),
                       static_cast<double*>(ws));  // env blocks of the workspace: (V,16) elastic preconditioner blocks per env
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail_hip(e, "fem_newton_lds_kernel");
  }
  if (resident) *resident = false;
  c->last_resident = 0;
  if (c->dev.indenters && c->dev.im_nt > 0) {
    set_error("FEM Newton: a mesh indenter needs the CU-resident Newton kernel (mesh with <= 512 vertices whose state fits the CU's 160 KB of "
              "LDS - this one needs %zu bytes; TACEX_FEM_NEWTON_LDS != 0); the streaming kernel of larger meshes handles analytic indenters "
              "(barrier, step bound, friction) only", lds);
    return 2;
  }
  // x, p and the H.p accumulators in LDS when the summation order is free (atomic mode) and they fit (9 V doubles: ~2 200 vertices)
  static const int stream_lds = getenv("TACEX_FEM_STREAM_LDS") ? atoi(getenv("TACEX_FEM_STREAM_LDS")) : 2;  // A/B hook: 0 none, 1 x / p / accumulators, 2 all PCG vectors
  const bool lds_all = stream_lds != 1 && (size_t)21 * V * sizeof(double) <= 160 * 1024;  // (TACEX_FEM_STREAM_LDS=1: x, p, accumulators only)
  const size_t lds_s = (size_t)(lds_all ? 21 : 9) * V * sizeof(double);
  const bool lds_sweep = stream_lds != 0 && atom && lds_s <= 160 * 1024;
  if (lds_sweep) {
    static size_t granted_s[64] = {};
    hipError_t es = hipSetDevice(c->device);
    if (es == hipSuccess) es = ensure_dynamic_lds(reinterpret_cast<const void*>(fem_newton_kernel), lds_s, granted_s);
    if (es != hipSuccess) return fail_hip(es, "hipFuncSetAttribute(fem_newton_kernel)");
  }
  hipLaunchKernelGGL(fem_newton_kernel, dim3(B), dim3(512), lds_sweep ? lds_s : 0, st, c->dev, x, xt, cons, aim, stats, static_cast<double*>(ws), pcg_max_iter,
                     pcg_tol_rate, ls_max_iter, dx_dev, dx_tol, step_info, stream_accumulate ? 1 : 0, fric ? xprev : nullptr, fric ? disp : nullptr,
                     c->fric_lag_mode == 1 ? 1 : 0, lds_sweep ? (lds_all ? 2 : 1) : 0);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_newton_kernel");
}

int tacex_fem_newton_step(tacex_fem_ctx* c, double* x, const double* xt, const uint8_t* cons, const double* aim,
                          double* stats, void* ws, int B, int pcg_max_iter, double pcg_tol_rate, int ls_max_iter,
                          void* stream) {
  if (!c || !x || !xt || !stats || !ws) { set_error("tacex_fem_newton_step: null argument"); return 2; }
  if ((cons == nullptr) != (aim == nullptr)) { set_error("tacex_fem_newton_step: constrained_dev and aim_dev go together"); return 2; }
  if (pcg_max_iter < 1 || ls_max_iter < 0 || !(pcg_tol_rate > 0.0)) { set_error("tacex_fem_newton_step: bad solver parameters"); return 2; }
  if (B <= 0) return 0;
  return launch_newton(c, x, xt, cons, aim, stats, ws, B, pcg_max_iter, pcg_tol_rate, ls_max_iter, c->dx_dev, c->dx_tol, 1, nullptr,
                       (hipStream_t)stream, nullptr);
}

int tacex_fem_step(tacex_fem_ctx* c, double* x, double* v, double* xt, const uint8_t* cons, const double* aim, double* stats,
                   double* step_info, void* ws, int B, const double gravity[3], int max_newton, double velocity_tol, int pcg_max_iter,
                   double pcg_tol_rate, int ls_max_iter, void* stream) {
  if (!c || !x || !v || !xt || !stats || !step_info || !ws || !gravity) { set_error("tacex_fem_step: null argument"); return 2; }
  if ((cons == nullptr) != (aim == nullptr)) { set_error("tacex_fem_step: constrained_dev and aim_dev go together"); return 2; }
  if (max_newton < 1 || pcg_max_iter < 1 || ls_max_iter < 0 || !(pcg_tol_rate > 0.0) || !(velocity_tol >= 0.0)) {
    set_error("tacex_fem_step: bad solver parameters");
    return 2;
  }
  if (B <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int V = c->dev.V;
  const size_t n3 = (size_t)B * V * 3;
  double* xprev = static_cast<double*>(ws) + (size_t)B * newton_ws_doubles(V, c->dev.T);
  double* dx = xprev + n3;  // (B,) max |d| per env: the device-side convergence state of this time step
  double* disp = dx + B;    // (B,3) indenter displacement since the previous step | (B,3) indenter position of the previous step
  double* ind_prev = disp + (size_t)3 * B;
  const double* ind = c->dev.indenters;
  const double dt = c->dev.dt, tol = velocity_tol * dt;
  hipLaunchKernelGGL(fem_predict_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, x, v, xt, xprev, dx, n3, B, dt, gravity[0],
                     gravity[1], gravity[2], ind, ind_prev, disp, (ind && c->ind_prev_ws == ws && c->ind_prev_B == B) ? 1 : 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "fem_predict_kernel");
  // heaviest envs first (see fem_newton_lds_kernel): worth it once a CU gets more than one env; TACEX_FEM_ORDER=0 = index order (A/B)
  static const int use_order = getenv("TACEX_FEM_ORDER") ? atoi(getenv("TACEX_FEM_ORDER")) : 1;
  int* env_order = nullptr;
  if (use_order && B > 1) {
    env_order = reinterpret_cast<int*>(ind_prev + (size_t)3 * B + 1);  // behind the (B,3) previous indenter positions (tacex_fem_workspace_bytes)
    hipLaunchKernelGGL(fem_env_order_kernel, dim3(1), dim3(1024), 0, st, step_info, B, env_order);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "fem_env_order_kernel");
  }
  bool resident = false;
  if (int rc = launch_newton(c, x, xt, cons, aim, stats, ws, B, pcg_max_iter, pcg_tol_rate, ls_max_iter, dx, tol, max_newton, step_info, st,
                             &resident, xprev, disp, env_order))
    return rc;
  if (!resident) {
    // streaming fallback: one launch per Newton iteration on a FIXED schedule; converged envs return at once (dx protocol), so the
    // launches past convergence cost microseconds and nothing is read back.  The first launch (above) SET the env's step_info row,
    // these add their iteration / PCG counts and OR their flags into it (penetration, dead line search: check_step() sees them).
    for (int it = 1; it < max_newton; ++it)
      if (int rc = launch_newton(c, x, xt, cons, aim, stats, ws, B, pcg_max_iter, pcg_tol_rate, ls_max_iter, dx, tol, 1, step_info, st, nullptr,
                                 xprev, disp, nullptr, true))  // (xprev / disp: the friction terms of the later iterations; the lag stays the first launch's)
        return rc;
  }
  hipLaunchKernelGGL(fem_velocity_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, x, xprev, v, n3, 1.0 / dt, ind, ind_prev, B);
  if (ind) { c->ind_prev_ws = ws; c->ind_prev_B = B; }
  e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_velocity_kernel");
}

// ---- the reference's UIPC scene: free affine-body ball + ground (fem_ball.h) ---------------------------------------------------------
int tacex_fem_set_affine_body(tacex_fem_ctx* c, int num_verts, const double* verts_host, int num_tris, const int32_t* tris_host, double density,
                              double kappa, const double* pad_vertex_area_host, int num_pad_tris, const int32_t* pad_tris_host, double d_hat,
                              double stiffness, double ground_height, int enable_ground, int kinematic) {
  if (!c) { set_error("tacex_fem_set_affine_body: null context"); return 2; }
  bool synced = false;
  BallDev& bd = c->ball;
  fem_release(c, bd.Y, &synced); fem_release(c, bd.tri, &synced); fem_release(c, bd.area, &synced);
  fem_release(c, bd.ptri, &synced); fem_release(c, bd.psv, &synced); fem_release(c, bd.parea, &synced);
  fem_release(c, bd.pedge, &synced); fem_release(c, bd.pearea, &synced); fem_release(c, bd.pelen2, &synced);
  fem_release(c, bd.bedge, &synced); fem_release(c, bd.bearea, &synced); fem_release(c, bd.belen2, &synced);
  bd = BallDev{};
  c->ball_last_ws = nullptr; c->ball_last_B = 0;
  if (num_verts == 0) return 0;  // remove the body
  if (!verts_host || !tris_host || !pad_vertex_area_host || !pad_tris_host || num_verts < 4 || num_tris < 4 || num_pad_tris < 1) {
    set_error("tacex_fem_set_affine_body: null / too small mesh argument");
    return 2;
  }
  if (num_verts >= 32768 || num_tris >= 32768 || num_pad_tris >= 32768 || c->dev.V >= 32768) {
    set_error("tacex_fem_set_affine_body: the pair list packs indices into 15 bits (meshes of < 32768 vertices / triangles)");
    return 2;
  }
  if (!(density > 0.0) || !(kappa > 0.0) || !(d_hat > 0.0) || !(stiffness > 0.0)) { set_error("tacex_fem_set_affine_body: bad material / contact parameters"); return 2; }
  const int V = c->dev.V;
  std::vector<double> Y((size_t)num_verts * 4), area(num_verts, 0.0), parea(pad_vertex_area_host, pad_vertex_area_host + V);
  std::vector<int> tri(tris_host, tris_host + (size_t)num_tris * 3), ptri(pad_tris_host, pad_tris_host + (size_t)num_pad_tris * 3), psv;
  for (int t : tri) if (t < 0 || t >= num_verts) { set_error("tacex_fem_set_affine_body: triangle index out of range"); return 2; }
  for (int t : ptri) if (t < 0 || t >= V) { set_error("tacex_fem_set_affine_body: pad triangle index out of range"); return 2; }
  for (int k = 0; k < num_verts; ++k) { Y[k * 4] = 1.0; for (int i = 0; i < 3; ++i) Y[k * 4 + 1 + i] = verts_host[k * 3 + i]; }
  // moments of the closed surface (signed tetrahedra against the origin), vertex areas - oracle/abd_oracle.py AffineBody
  double S[16] = {}, vol = 0.0;
  for (int t = 0; t < num_tris; ++t) {
    const double* a = verts_host + (size_t)tri[t * 3] * 3; const double* b = verts_host + (size_t)tri[t * 3 + 1] * 3; const double* cc = verts_host + (size_t)tri[t * 3 + 2] * 3;
    const double bc[3] = {b[1] * cc[2] - b[2] * cc[1], b[2] * cc[0] - b[0] * cc[2], b[0] * cc[1] - b[1] * cc[0]};
    const double v6 = (a[0] * bc[0] + a[1] * bc[1] + a[2] * bc[2]) / 6.0;
    vol += v6;
    S[0] += v6;
    const double sm[3] = {a[0] + b[0] + cc[0], a[1] + b[1] + cc[1], a[2] + b[2] + cc[2]};
    for (int i = 0; i < 3; ++i) {
      S[1 + i] += v6 * sm[i] / 4.0;
      for (int j = 0; j < 3; ++j) S[(1 + i) * 4 + 1 + j] += v6 / 20.0 * (a[i] * a[j] + b[i] * b[j] + cc[i] * cc[j] + sm[i] * sm[j]);
    }
    const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]};
    const double cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double ta = 0.5 * sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    for (int k = 0; k < 3; ++k) area[tri[t * 3 + k]] += ta / 3.0;
  }
  if (!(vol > 0.0)) { set_error("tacex_fem_set_affine_body: the surface triangles must be oriented outward (enclosed volume %g)", vol); return 2; }
  for (int i = 1; i < 4; ++i) S[i * 4] = S[i];
  for (int k = 0; k < 16; ++k) bd.S[k] = density * S[k];
  for (int v = 0; v < V; ++v) if (parea[v] > 0.0) psv.push_back(v);
  bd.nv = num_verts; bd.nt = num_tris; bd.npt = num_pad_tris; bd.nsv = (int)psv.size();
  bd.kv = kappa * vol; bd.gh = ground_height; bd.dhat = d_hat; bd.kappa = stiffness; bd.ground = enable_ground ? 1 : 0; bd.kinematic = kinematic ? 1 : 0;
  if (int rc = fem_upload(c, Y, &bd.Y)) return rc;
  if (int rc = fem_upload(c, tri, &bd.tri)) return rc;
  if (int rc = fem_upload(c, area, &bd.area)) return rc;
  if (int rc = fem_upload(c, ptri, &bd.ptri)) return rc;
  if (int rc = fem_upload(c, psv, &bd.psv)) return rc;
  if (int rc = fem_upload(c, parea, &bd.parea)) return rc;
  // edge-edge pairs: the unique edges of both surfaces (oracle/abd_oracle.py surface_edges); the pad's rest positions come back from the device
  {
    std::vector<double> rest((size_t)3 * V);
    hipError_t e = hipMemcpy(rest.data(), c->rest, rest.size() * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpy(pad rest positions)");
    std::vector<int> pe, be;
    std::vector<double> pa, pl, ba, bl;
    surface_edges(ptri, rest.data(), pe, pa, pl);
    surface_edges(tri, verts_host, be, ba, bl);
    if (pe.size() / 2 >= 32768 || be.size() / 2 >= 32768) { set_error("tacex_fem_set_affine_body: more than 32767 surface edges"); return 2; }
    bd.npe = (int)(pe.size() / 2); bd.nbe = (int)(be.size() / 2); bd.ee = 1;
    if (int rc = fem_upload(c, pe, &bd.pedge)) return rc;
    if (int rc = fem_upload(c, pa, &bd.pearea)) return rc;
    if (int rc = fem_upload(c, pl, &bd.pelen2)) return rc;
    if (int rc = fem_upload(c, be, &bd.bedge)) return rc;
    if (int rc = fem_upload(c, ba, &bd.bearea)) return rc;
    if (int rc = fem_upload(c, bl, &bd.belen2)) return rc;
  }
  return 0;
}

int tacex_fem_set_edge_edge(tacex_fem_ctx* c, int enable) {
  if (!c || c->ball.nv == 0) { set_error("tacex_fem_set_edge_edge: no affine body set"); return 2; }
  c->ball.ee = enable ? 1 : 0;
  return 0;
}

size_t tacex_fem_ball_workspace_bytes(const tacex_fem_ctx* c, int B) {
  if (!c || B <= 0 || c->ball.nv == 0) return 0;
  // env blocks | x_prev (B,V,3) | q_prev (B,12) | x~ (B,V,3) | q~ (B,12) | q at the end of the previous step (B,12) | env launch order (B int32)
  // | elastic preconditioner blocks (B,V,16)
  return ((size_t)B * ball_ws_doubles(c->dev.V, c->dev.T, c->ball.nv, c->ball.nt) + (size_t)B * 6 * c->dev.V + (size_t)B * 36 + 8 + ((size_t)B + 1) / 2 + 2 +
          (size_t)B * 16 * c->dev.V) * sizeof(double);
}

int tacex_fem_ball_moments(const tacex_fem_ctx* c, double moments_out[16], double* kappa_vol_out) {
  if (!c || c->ball.nv == 0 || !moments_out) { set_error("tacex_fem_ball_moments: no affine body set"); return 2; }
  for (int k = 0; k < 16; ++k) moments_out[k] = c->ball.S[k];
  if (kappa_vol_out) *kappa_vol_out = c->ball.kv;
  return 0;
}

// threads per env of fem_ball_newton_kernel: 512 (two waves per SIMD at 256 VGPRs: 358 of them spilled) or 256 (one wave per SIMD with the whole
// register file); TACEX_BALL_NT picks (A/B, profiles/r06_experiments.md section 13)
static int ball_threads() {
  static const int nt = getenv("TACEX_BALL_NT") ? atoi(getenv("TACEX_BALL_NT")) : 512;
  return nt == 256 ? 256 : 512;
}

static int ball_lds_ok(tacex_fem_ctx* c, const char* who) {
  const size_t lds = ball_lds_bytes(c->dev.V);
  if (lds > 160 * 1024) { set_error("%s: pad of %d vertices (x, p, accumulators and chain factors of one env must fit a CU's 160 KB of LDS)", who, c->dev.V); return 2; }
  static size_t granted[2][64] = {};
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess)
    e = ball_threads() == 256 ? ensure_dynamic_lds(reinterpret_cast<const void*>(fem_ball_newton_kernel<256>), lds, granted[0])
                              : ensure_dynamic_lds(reinterpret_cast<const void*>(fem_ball_newton_kernel<512>), lds, granted[1]);
  return e == hipSuccess ? 0 : fail_hip(e, "hipFuncSetAttribute(fem_ball_newton_kernel)");
}

static int ball_args_ok(tacex_fem_ctx* c, const void* x, const void* q, const void* ws, const uint8_t* cons, const double* aim, const char* who) {
  if (!c || !x || !q || !ws) { set_error("%s: null argument", who); return 2; }
  if (c->ball.nv == 0) { set_error("%s: no affine body (tacex_fem_set_affine_body)", who); return 2; }
  if ((size_t)9 * c->dev.V > (size_t)12 * c->dev.T) { set_error("%s: mesh with more than 4/3 vertices per tet (the lagged blocks share a per-tet array)", who); return 2; }
  if ((cons == nullptr) != (aim == nullptr)) { set_error("%s: constrained_dev and aim_dev go together", who); return 2; }
  return 0;
}

int tacex_fem_ball_terms(tacex_fem_ctx* c, const double* x, const double* xt, const double* q, const double* qt, const uint8_t* cons, const double* aim,
                         const double* x_prev, const double* q_prev, double* energy, double* grad, double* step_info, void* ws, int B, void* stream) {
  if (int rc = ball_args_ok(c, x, q, ws, cons, aim, "tacex_fem_ball_terms")) return rc;
  if (!xt || !qt) { set_error("tacex_fem_ball_terms: null argument"); return 2; }
  if (B <= 0) return 0;
  if (int rc = ball_lds_ok(c, "tacex_fem_ball_terms")) return rc;
  auto kern = ball_threads() == 256 ? fem_ball_newton_kernel<256> : fem_ball_newton_kernel<512>;
  hipLaunchKernelGGL(kern, dim3(B), dim3(ball_threads()), ball_lds_bytes(c->dev.V), (hipStream_t)stream, c->dev, c->ball, const_cast<double*>(x), xt,
                     const_cast<double*>(q), qt, cons, aim, static_cast<double*>(ws), 1, 1.0, 0, 1, 0.0, 0.0, step_info, 1, energy, grad,
                     (x_prev && q_prev) ? x_prev : nullptr, (x_prev && q_prev) ? q_prev : nullptr, static_cast<const int*>(nullptr),
                     static_cast<const double*>(nullptr));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_ball_newton_kernel(terms)");
}

int tacex_fem_ball_step(tacex_fem_ctx* c, double* x, double* v, double* q, double* qv, const uint8_t* cons, const double* aim, double* step_info,
                        void* ws, int B, const double gravity[3], int max_newton, double velocity_tol, double transrate_tol, int pcg_max_iter,
                        double pcg_tol_rate, int ls_max_iter, void* stream) {
  if (int rc = ball_args_ok(c, x, q, ws, cons, aim, "tacex_fem_ball_step")) return rc;
  if (!v || !qv || !step_info || !gravity) { set_error("tacex_fem_ball_step: null argument"); return 2; }
  if (max_newton < 1 || pcg_max_iter < 1 || ls_max_iter < 0 || !(pcg_tol_rate > 0.0) || !(velocity_tol >= 0.0) || !(transrate_tol >= 0.0)) {
    set_error("tacex_fem_ball_step: bad solver parameters");
    return 2;
  }
  if (B <= 0) return 0;
  static const int ball_coarse_off = getenv("TACEX_BALL_COARSE") ? (atoi(getenv("TACEX_BALL_COARSE")) == 0) : 0;  // A/B hook: block Jacobi alone on the pad rows
  hipStream_t st = (hipStream_t)stream;
  const int V = c->dev.V;
  const size_t n3 = (size_t)B * V * 3;
  double* xprev = static_cast<double*>(ws) + (size_t)B * ball_ws_doubles(V, c->dev.T, c->ball.nv, c->ball.nt);
  double* qprev = xprev + n3;
  double* xt = qprev + (size_t)B * 12;
  double* qt = xt + n3;
  // A KINEMATIC body is moved by the caller between steps: friction then slides relative to where the body stood at the END of the previous
  // step (kept here; the first step with this workspace sees no body motion), like the analytic indenters' displacement in tacex_fem_step
  double* qlast = qt + (size_t)B * 12;
  const bool have_last = c->ball.kinematic && c->ball_last_ws == ws && c->ball_last_B == B;
  const double dt = c->dev.dt;
  hipLaunchKernelGGL(fem_predict_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, x, v, xt, xprev, static_cast<double*>(nullptr), n3, B, dt,
                     gravity[0], gravity[1], gravity[2], static_cast<const double*>(nullptr), static_cast<double*>(nullptr), static_cast<double*>(nullptr), 0);
  hipLaunchKernelGGL(fem_ball_predict_kernel, dim3((unsigned)((B * 12 + 255) / 256)), dim3(256), 0, st, q, qv, qt, qprev, B, dt, gravity[0], gravity[1],
                     gravity[2]);
  int* env_order = nullptr;
  static const int use_order = getenv("TACEX_FEM_ORDER") ? atoi(getenv("TACEX_FEM_ORDER")) : 1;
  if (use_order && B > 256) {  // (step_info still holds the previous step's counts; a first step sorts zeros = index order)
    env_order = reinterpret_cast<int*>(qlast + (size_t)B * 12 + 1);
    hipLaunchKernelGGL(fem_env_order_kernel, dim3(1), dim3(1024), 0, st, step_info, B, env_order);
  }
  if (int rc = ball_lds_ok(c, "tacex_fem_ball_step")) return rc;
  // elastic preconditioner blocks D | E of every env at the state the step starts from (what fem_newton_lds_kernel's launch does too)
  double* blk = reinterpret_cast<double*>(reinterpret_cast<char*>(qlast + (size_t)B * 12) + (((size_t)B + 1) / 2 + 2) * sizeof(double));
  {
    static size_t granted_a[2][64] = {};
    const bool asm_atom = !c->deterministic;
    const size_t lds_a = ((size_t)3 * V + (asm_atom ? (size_t)15 * V : 0)) * sizeof(double);
    auto ka = asm_atom ? fem_assemble_blocks_kernel<true> : fem_assemble_blocks_kernel<false>;
    hipError_t ea = hipSetDevice(c->device);
    if (ea == hipSuccess) ea = ensure_dynamic_lds(reinterpret_cast<const void*>(ka), lds_a, granted_a[asm_atom ? 1 : 0]);
    if (ea != hipSuccess) return fail_hip(ea, "hipFuncSetAttribute(fem_assemble_blocks_kernel)");
    hipLaunchKernelGGL(ka, dim3(B), dim3(512), lds_a, st, c->dev_nwt, x, blk, static_cast<const double*>(nullptr), 0.0);
  }
  auto kern = ball_threads() == 256 ? fem_ball_newton_kernel<256> : fem_ball_newton_kernel<512>;
  hipLaunchKernelGGL(kern, dim3(B), dim3(ball_threads()), ball_lds_bytes(c->dev.V), st, c->dev, c->ball, x, xt, q, qt, cons, aim, static_cast<double*>(ws), pcg_max_iter,
                     pcg_tol_rate, ls_max_iter, max_newton, velocity_tol * dt, transrate_tol * dt, step_info, (ball_coarse_off ? 2 : 0) | ((c->ls_refine & 15) << 8),
                     static_cast<double*>(nullptr), static_cast<double*>(nullptr), static_cast<const double*>(xprev),
                     static_cast<const double*>(have_last ? qlast : qprev), static_cast<const int*>(env_order), static_cast<const double*>(blk));
  hipLaunchKernelGGL(fem_velocity_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, x, xprev, v, n3, 1.0 / dt, static_cast<const double*>(nullptr),
                     static_cast<double*>(nullptr), B);
  hipLaunchKernelGGL(fem_ball_velocity_kernel, dim3((unsigned)((B * 12 + 255) / 256)), dim3(256), 0, st, q, qprev, qv, B, 1.0 / dt);
  if (c->ball.kinematic) {
    (void)hipMemcpyAsync(qlast, q, (size_t)B * 12 * sizeof(double), hipMemcpyDeviceToDevice, st);
    c->ball_last_ws = ws; c->ball_last_B = B;
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "tacex_fem_ball_step");
}

int tacex_fem_reset_envs(tacex_fem_ctx* c, const int32_t* env_ids, int num_reset, const double* positions, double* x, double* v,
                         double* step_info, void* ws, int B, void* stream) {
  if (!c || !x || !v) { set_error("tacex_fem_reset_envs: null argument"); return 2; }
  if (B <= 0 || num_reset == 0) return 0;
  if (num_reset < 0 || (!env_ids && num_reset != B)) { set_error("tacex_fem_reset_envs: without env_ids, num_reset must equal num_envs"); return 2; }
  const int V = c->dev.V;
  // the previous indenter positions live in the workspace the last tacex_fem_step ran with (see tacex_fem_step): another workspace or
  // env count means the next step has no previous positions at all
  double* ind_prev = nullptr;
  if (ws && c->dev.indenters && c->ind_prev_ws == ws && c->ind_prev_B == B)
    ind_prev = static_cast<double*>(ws) + (size_t)B * newton_ws_doubles(V, c->dev.T) + (size_t)B * V * 3 + B + (size_t)3 * B;
  hipLaunchKernelGGL(fem_reset_envs_kernel, dim3(num_reset), dim3(256), 0, (hipStream_t)stream, env_ids, positions, c->rest, x, v, step_info,
                     ind_prev, V, B);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_reset_envs_kernel");
}

int tacex_fem_set_attachment_targets(const float* body_pos, const float* body_quat, const float* offsets, const int32_t* idx,
                                     double* aim, uint8_t* constrained, double* aim_compact, int B, int A, int V, void* stream) {
  if (!body_pos || !body_quat || !offsets || !idx || !aim || !constrained) { set_error("tacex_fem_set_attachment_targets: null argument"); return 2; }
  if (B <= 0 || A <= 0) return 0;
  hipLaunchKernelGGL(fem_attachment_aim_kernel, dim3((A + 127) / 128, B), dim3(128), 0, (hipStream_t)stream, body_pos, body_quat,
                     offsets, idx, aim, constrained, aim_compact, A, V);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_attachment_aim_kernel");
}

int tacex_fem_marker_uv(const double* pos, const int32_t* tri, const double* wgt, double fx, double fy, double cx,
                        double cy, double* uv, int B, int Vs, int M, void* stream) {
  if (!pos || !tri || !wgt || !uv) { set_error("tacex_fem_marker_uv: null argument"); return 2; }
  if (B <= 0 || M <= 0) return 0;
  hipLaunchKernelGGL(fem_marker_uv_kernel, dim3((M + 127) / 128, B), dim3(128), 0, (hipStream_t)stream, pos, tri, wgt, fx, fy,
                     cx, cy, uv, Vs, M);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_marker_uv_kernel");
}

int tacex_fem_marker_flow(const double* x, const int64_t* surf_ids, const double* cam_pos, const double* cam_rot_inv, const int32_t* tri,
                          const double* wgt, double fx, double fy, double cx, double cy, const double* init_uv, const int64_t* select,
                          double normalize_div, double* curr_uv, double* flow, float* flow_f32, int B, int V, int M, int K, void* stream) {
  if (!x || !surf_ids || !cam_pos || !cam_rot_inv || !tri || !wgt || !init_uv || !select || (!flow && !flow_f32)) {
    set_error("tacex_fem_marker_flow: null argument");
    return 2;
  }
  if (B <= 0 || M <= 0 || K <= 0) return 0;
  if ((size_t)M * 2 * sizeof(double) > 64 * 1024) { set_error("tacex_fem_marker_flow: %d markers (the env's projections are staged in 64 KB of LDS)", M); return 2; }
  hipLaunchKernelGGL(fem_marker_flow_kernel, dim3(B), dim3(256), (size_t)M * 2 * sizeof(double), (hipStream_t)stream, x,
                     reinterpret_cast<const long long*>(surf_ids), cam_pos, cam_rot_inv, tri, wgt, fx, fy, cx, cy, init_uv,
                     reinterpret_cast<const long long*>(select), normalize_div, curr_uv, flow, flow_f32, V, M, K);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "fem_marker_flow_kernel");
}

}  // extern "C"
