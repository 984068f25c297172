// rt_shell.hip -- the consumer of the Monte Carlo curve: the velocity of a radiation-driven gas shell,
// integrated for every H II region of a galaxy at once (one thread per region).
//
// Reference semantics (file:line into the reference checkout):
//   get_tau_IR          MC code.py:781-807    infrared optical depth of the shell
//   GetMC               MC code.py:843-863    momentum transfer function at (tau, log age): the curve runMC wrote
//   ODEs                MC code.py:905-955    dv/dr and dt/dr of the spherical shell, with or without the light and
//                                             gravity of the galaxy's other regions (includeWholeGalaxy, :919-920, :940-941)
//   GetRegionVelocity   MC code.py:1433-1472  integration from h_g to 100 H_old, v(h_g) = v_0, t(h_g) = t_0
// The reference integrates one region at a time with scipy's Radau (max_step = 1 pc); here an embedded
// Dormand-Prince 5(4) pair with the same step cap runs one region per lane.  The system is not stiff;
// what makes it delicate is dv/dr ~ 1/v when a shell stalls, which step-size control handles.
// With the galaxy's other regions every right-hand side holds two sums over all rows of gal_data: then one
// wave integrates one region, all 64 lanes carry the same state and share the sums.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "rt_kernels.hpp"

namespace rt {
namespace {

// np.interp for an ascending table: clamped ends, slope form.
__device__ __forceinline__ double interp_clamped(const double* __restrict__ x, const double* __restrict__ y, int n,
                                                 double q) {
    if (q <= x[0]) return y[0];
    if (q >= x[n - 1]) return y[n - 1];
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (x[mid] <= q)
            lo = mid;
        else
            hi = mid;
    }
    const double slope = (y[hi] - y[lo]) / (x[hi] - x[lo]);
    return slope * (q - x[lo]) + y[lo];
}

// index i with x[i] <= q < x[i+1], clipped to [0, n-2] (the cell scipy's linear interpn uses)
__device__ __forceinline__ int cell_of(const double* __restrict__ x, int n, double q) {
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (x[mid] <= q)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// One entry of the curve with the optically thin replacement below the cut-off (MC code.py:849-853).
__device__ __forceinline__ double mc_entry(const ShellModel& m, int i_tau, int i_age, double ratio) {
    const double tau = m.mc_tau[i_tau];
    if (m.tau_cutoff > 0.0 && tau * ratio < m.tau_cutoff) return 1.0 - exp(-tau * ratio);
    return m.mc_val[(size_t)i_tau * m.n_age + i_age];
}

// GetMC: np.interp along tau when log_age is one of the tabulated ages, bilinear otherwise.  Arguments
// outside the table are clamped (the reference's interpn raises there).
__device__ double get_mc(const ShellModel& m, double tau, double log_age) {
    const double ratio = m.tau_cutoff > 0.0 ? interp_clamped(m.l_age, m.ratio_val, m.n_lage, log_age) : 0.0;
    int exact = -1;
    for (int k = 0; k < m.n_age; ++k)
        if (m.mc_age[k] == log_age) exact = k;
    if (exact >= 0) {
        if (tau <= m.mc_tau[0]) return mc_entry(m, 0, exact, ratio);
        if (tau >= m.mc_tau[m.n_tau - 1]) return mc_entry(m, m.n_tau - 1, exact, ratio);
        const int i = cell_of(m.mc_tau, m.n_tau, tau);
        const double y0 = mc_entry(m, i, exact, ratio), y1 = mc_entry(m, i + 1, exact, ratio);
        const double slope = (y1 - y0) / (m.mc_tau[i + 1] - m.mc_tau[i]);
        return slope * (tau - m.mc_tau[i]) + y0;
    }
    const double tq = fmin(fmax(tau, m.mc_tau[0]), m.mc_tau[m.n_tau - 1]);
    const double aq = fmin(fmax(log_age, m.mc_age[0]), m.mc_age[m.n_age - 1]);
    const int i = cell_of(m.mc_tau, m.n_tau, tq), k = cell_of(m.mc_age, m.n_age, aq);
    const double y0 = (tq - m.mc_tau[i]) / (m.mc_tau[i + 1] - m.mc_tau[i]);
    const double y1 = (aq - m.mc_age[k]) / (m.mc_age[k + 1] - m.mc_age[k]);
    return (1.0 - y0) * (1.0 - y1) * mc_entry(m, i, k, ratio) + (1.0 - y0) * y1 * mc_entry(m, i, k + 1, ratio) +
           y0 * (1.0 - y1) * mc_entry(m, i + 1, k, ratio) + y0 * y1 * mc_entry(m, i + 1, k + 1, ratio);
}

// get_tau_IR, MC code.py:790-803
__device__ __forceinline__ double tau_ir(const ShellModel& m, double r, double l_bol, double sigma_g) {
    const double t_eff = pow(l_bol / (4.0 * 3.141592653589793 * (r * r) * m.sigma_sb), 0.25);
    const double tau_r = 2.4e-4 * (t_eff * t_eff) * sigma_g;
    if (tau_r > 1.0) {
        const double t4 = (t_eff * t_eff) * (t_eff * t_eff);
        const double temp = pow(3.0 / 4.0 * (tau_r + 2.0 / 3.0) * t4, 0.25);
        return 2.4e-4 * (temp * temp) * sigma_g;
    }
    return tau_r;
}

struct Region {
    double m_g, m_new, m_old, m_co_hi, h_old, h_g_i, tau_i, l_bol, sigma_g;
};

__device__ __forceinline__ Region load_region(const ShellRegions& rg, int i) {
    return {rg.m_g[i], rg.m_new[i], rg.m_old[i], rg.m_co_hi[i], rg.h_old[i], rg.h_g_i[i], rg.tau_i[i], rg.l_bol[i],
            rg.sigma_g[i]};
}

// sin(arctan(x)) for x >= 0 or NaN: x / sqrt(1 + x^2), which rounds to 1 from x = 2^27 on (x = +inf: dist_to_ref = 0)
__device__ __forceinline__ double sin_of_arctan(double x) { return x >= 134217728.0 ? 1.0 : x / sqrt(1.0 + x * x); }

// Sum over the wave; every lane returns the same bits (each butterfly step adds the same two numbers on both sides).
__device__ __forceinline__ double wave_sum(double a) {
    for (int s = 32; s >= 1; s >>= 1) a += __shfl_xor(a, s, 64);
    return a;
}

// The two sums of ODEs over gal_data (MC code.py:920, :941), formulas and units as written there.  Called by all 64
// lanes of a wave with the same arguments; lane l takes rows l, l + 64, ...
struct GalaxySums {
    double light;    // sum over all rows of LHa_ratio_f * sin(arctan(r / dist_to_ref / cm_pc))
    double gravity;  // sum over rows != self of Mass_tot / ((dist_to_ref / cm_pc)^2 + r^2) * sin(arctan(...))
};
__device__ GalaxySums galaxy_sums(const ShellGalaxy& g, int view, int self, double r) {
    const double* __restrict__ lha = g.lha_ratio_f + (size_t)view * g.n_gal;
    const double* __restrict__ dist = g.dist_to_ref + (size_t)view * g.n_gal;
    double light = 0.0, gravity = 0.0;
    for (int j = (int)(threadIdx.x & 63u); j < g.n_gal; j += 64) {
        const double d = dist[j];
        const double toward = sin_of_arctan(r / d / g.cm_pc);
        light += lha[j] * toward;
        const double dd = d / g.cm_pc;
        if (j != self) gravity += g.mass_tot[j] / (dd * dd + r * r) * toward;
    }
    return {wave_sum(light), wave_sum(gravity)};
}

// ODEs, MC code.py:913-955.  GAL: includeWholeGalaxy, `view` / `self` are the region's entries of view_of_region / self_row.
template <bool GAL>
__device__ void rhs(const ShellModel& m, const ShellGalaxy& g, int view, int self, const Region& q, double r, double v,
                    double t, double& dvdr, double& dtdr) {
    const double new_time = log10(t);
    double l_bol = q.l_bol;
    GalaxySums gs = {0.0, 0.0};
    if constexpr (GAL) {
        gs = galaxy_sums(g, view, self, r);
        l_bol = q.l_bol * (1.0 - exp(-q.tau_i * g.ratio_rp_lambda_rp) + gs.light);
    }
    const double s = q.h_g_i / r;
    const double tau_new = (s * s) * q.tau_i;
    const double sigma_now = (s * s) * q.sigma_g;
    double log_age = m.time_slice;
    if (m.aging) {
        l_bol *= interp_clamped(m.l_age, m.l_val, m.n_lage, new_time) / m.l_norm;
        log_age = new_time;
    }
    const double momentum = get_mc(m, m.static_momentum ? q.tau_i : tau_new, log_age) + tau_ir(m, r, l_bol, sigma_now);
    const double a = fmin(r, m.h_hi_cm) / q.h_g_i, b = fmin(r, q.h_old) / q.h_g_i;
    const double mass = q.m_g + q.m_co_hi * (a * a * a) + q.m_new + q.m_old * (b * b * b);
    dvdr = (-m.G * mass / (r * r) + momentum * l_bol / (m.c * q.m_g)) / v;
    if constexpr (GAL) dvdr -= m.G * gs.gravity / v;
    dtdr = 1.0 / (v * m.year_s);
    if (r > m.r_escape) dvdr = -__builtin_inf();
}

// GAL: one wave (= one workgroup of 64) per point, lane 0 stores; else one point per lane.
template <bool GAL>
__global__ void shell_rhs_kernel(const ShellModel m, const ShellRegions rg, const ShellGalaxy g, int n,
                                 const int32_t* __restrict__ region, const double* __restrict__ r,
                                 const double* __restrict__ v, const double* __restrict__ t, double* __restrict__ dvdr,
                                 double* __restrict__ dtdr) {
    const int i = GAL ? (int)blockIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const Region q = load_region(rg, region[i]);
    const int view = GAL ? g.view_of_region[region[i]] : 0, self = GAL ? g.self_row[region[i]] : -1;
    double a, b;
    rhs<GAL>(m, g, view, self, q, r[i], v[i], t[i], a, b);
    if (!GAL || threadIdx.x == 0) {
        dvdr[i] = a;
        dtdr[i] = b;
    }
}

// Dormand-Prince 5(4), one region per thread (GAL: per wave, every lane with the same state, so control flow is
// uniform and the shuffles of the row sums see all lanes; lane 0 stores).  Steps are capped by max_step and land
// exactly on the n_out output radii r0 + (k+1)/n_out (r_end - r0).
template <bool GAL>
__global__ void shell_ode_kernel(const ShellModel m, const ShellRegions rg, const ShellGalaxy g, const ShellControl ct,
                                 ShellResult out) {
    const int i = GAL ? (int)blockIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= rg.n) return;
    const bool writer = !GAL || threadIdx.x == 0;
    const Region q = load_region(rg, i);
    const int view = GAL ? g.view_of_region[i] : 0, self = GAL ? g.self_row[i] : -1;
    // beyond r_escape the reference sets dv/dr = -inf and its solver stops (MC code.py:951-953): integrate to there
    const double r0 = q.h_g_i, r_end = rg.r_end[i], r_stop = fmin(r_end, m.r_escape);
    double r = r0, v = ct.v0, t = ct.t0;
    double v_max = v, r_vmax = r, t_vmax = t;
    int status = RT_SHELL_REACHED_END, steps = 0, next_out = 0;
    double k1v, k1t;
    rhs<GAL>(m, g, view, self, q, r, v, t, k1v, k1t);
    double h = fmin(ct.max_step, 1e-3 * (r_end - r0));
    for (int k = 0; writer && k < ct.n_out; ++k) {
        out.v_out[(size_t)i * ct.n_out + k] = __builtin_nan("");
        out.t_out[(size_t)i * ct.n_out + k] = __builtin_nan("");
    }
    while (r < r_stop) {
        if (steps >= ct.max_steps) {
            status = RT_SHELL_STEP_LIMIT;
            break;
        }
        if (!(isfinite(k1v) && isfinite(k1t))) {
            status = RT_SHELL_ESCAPED;  // beyond r_escape the reference sets dv/dr = -inf and its solver stops
            break;
        }
        const double next_sample =
            ct.n_out > 0 && next_out < ct.n_out ? r0 + (double)(next_out + 1) / (double)ct.n_out * (r_end - r0) : r_end;
        const bool to_sample = next_sample <= r_stop;
        const double target = to_sample ? next_sample : r_stop;
        bool lands = false;
        double hs = fmin(h, ct.max_step);
        if (r + hs >= target || target - (r + hs) < 1e-12 * target) {
            hs = target - r;
            lands = true;
        }
        // stages
        double k2v, k2t, k3v, k3t, k4v, k4t, k5v, k5t, k6v, k6t, k7v, k7t;
        rhs<GAL>(m, g, view, self, q, r + hs * (1.0 / 5.0), v + hs * (1.0 / 5.0) * k1v, t + hs * (1.0 / 5.0) * k1t, k2v, k2t);
        rhs<GAL>(m, g, view, self, q, r + hs * (3.0 / 10.0), v + hs * (3.0 / 40.0 * k1v + 9.0 / 40.0 * k2v),
            t + hs * (3.0 / 40.0 * k1t + 9.0 / 40.0 * k2t), k3v, k3t);
        rhs<GAL>(m, g, view, self, q, r + hs * (4.0 / 5.0), v + hs * (44.0 / 45.0 * k1v - 56.0 / 15.0 * k2v + 32.0 / 9.0 * k3v),
            t + hs * (44.0 / 45.0 * k1t - 56.0 / 15.0 * k2t + 32.0 / 9.0 * k3t), k4v, k4t);
        rhs<GAL>(m, g, view, self, q, r + hs * (8.0 / 9.0),
            v + hs * (19372.0 / 6561.0 * k1v - 25360.0 / 2187.0 * k2v + 64448.0 / 6561.0 * k3v - 212.0 / 729.0 * k4v),
            t + hs * (19372.0 / 6561.0 * k1t - 25360.0 / 2187.0 * k2t + 64448.0 / 6561.0 * k3t - 212.0 / 729.0 * k4t),
            k5v, k5t);
        rhs<GAL>(m, g, view, self, q, r + hs,
            v + hs * (9017.0 / 3168.0 * k1v - 355.0 / 33.0 * k2v + 46732.0 / 5247.0 * k3v + 49.0 / 176.0 * k4v -
                      5103.0 / 18656.0 * k5v),
            t + hs * (9017.0 / 3168.0 * k1t - 355.0 / 33.0 * k2t + 46732.0 / 5247.0 * k3t + 49.0 / 176.0 * k4t -
                      5103.0 / 18656.0 * k5t),
            k6v, k6t);
        const double vn = v + hs * (35.0 / 384.0 * k1v + 500.0 / 1113.0 * k3v + 125.0 / 192.0 * k4v -
                                    2187.0 / 6784.0 * k5v + 11.0 / 84.0 * k6v);
        const double tn = t + hs * (35.0 / 384.0 * k1t + 500.0 / 1113.0 * k3t + 125.0 / 192.0 * k4t -
                                    2187.0 / 6784.0 * k5t + 11.0 / 84.0 * k6t);
        bool ok = isfinite(vn) && isfinite(tn) && vn > 0.0;
        double err = 2.0;
        if (ok) {
            rhs<GAL>(m, g, view, self, q, r + hs, vn, tn, k7v, k7t);
            const double ev = hs * (71.0 / 57600.0 * k1v - 71.0 / 16695.0 * k3v + 71.0 / 1920.0 * k4v -
                                    17253.0 / 339200.0 * k5v + 22.0 / 525.0 * k6v - 1.0 / 40.0 * k7v);
            const double et = hs * (71.0 / 57600.0 * k1t - 71.0 / 16695.0 * k3t + 71.0 / 1920.0 * k4t -
                                    17253.0 / 339200.0 * k5t + 22.0 / 525.0 * k6t - 1.0 / 40.0 * k7t);
            const double sv = ct.atol_v + ct.rtol * fmax(fabs(v), fabs(vn));
            const double st = ct.atol_t + ct.rtol * fmax(fabs(t), fabs(tn));
            err = sqrt(0.5 * ((ev / sv) * (ev / sv) + (et / st) * (et / st)));
            ok = isfinite(err) && err <= 1.0;
        }
        ++steps;
        if (ok) {
            r = lands ? target : r + hs;
            v = vn;
            t = tn;
            k1v = k7v;
            k1t = k7t;
            if (v > v_max) {
                v_max = v;
                r_vmax = r;
                t_vmax = t;
            }
            if (lands && to_sample && ct.n_out > 0 && next_out < ct.n_out) {
                if (writer) {
                    out.v_out[(size_t)i * ct.n_out + next_out] = v;
                    out.t_out[(size_t)i * ct.n_out + next_out] = t;
                }
                ++next_out;
            }
            const double grow = err > 0.0 ? fmin(5.0, 0.9 * pow(err, -0.2)) : 5.0;
            if (!lands || grow < 1.0) h = hs * fmax(0.2, grow);
            else h = fmax(h, hs * grow);
        } else {
            const double shrink = isfinite(err) && err > 1.0 ? fmax(0.1, 0.9 * pow(err, -0.2)) : 0.25;
            h = hs * fmin(shrink, 0.9);
            if (h < 1e-13 * r) {
                status = RT_SHELL_STALLED;  // v -> 0: dv/dr ~ 1/v cannot be followed further
                break;
            }
        }
    }
    if (status == RT_SHELL_REACHED_END && r_stop < r_end) status = RT_SHELL_ESCAPED;
    if (!writer) return;
    out.v_max[i] = v_max;
    out.r_vmax[i] = r_vmax;
    out.t_vmax[i] = t_vmax;
    out.r_last[i] = r;
    out.v_last[i] = v;
    out.t_last[i] = t;
    out.status[i] = status;
    out.n_steps[i] = steps;
}

}  // namespace

hipError_t launch_shell_rhs(const ShellModel& m, const ShellRegions& rg, const ShellGalaxy* gal, int n,
                            const int32_t* d_region, const double* d_r, const double* d_v, const double* d_t,
                            double* d_dvdr, double* d_dtdr, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (gal)
        shell_rhs_kernel<true><<<n, 64, 0, stream>>>(m, rg, *gal, n, d_region, d_r, d_v, d_t, d_dvdr, d_dtdr);
    else
        shell_rhs_kernel<false><<<(n + 63) / 64, 64, 0, stream>>>(m, rg, ShellGalaxy{}, n, d_region, d_r, d_v, d_t, d_dvdr, d_dtdr);
    return hipGetLastError();
}

hipError_t launch_shell_ode(const ShellModel& m, const ShellRegions& rg, const ShellGalaxy* gal, const ShellControl& ct,
                            const ShellResult& out, hipStream_t stream) {
    if (rg.n <= 0) return hipSuccess;
    if (gal)  // one wave per region
        shell_ode_kernel<true><<<rg.n, 64, 0, stream>>>(m, rg, *gal, ct, out);
    else  // 64 regions per workgroup: regions take very different numbers of steps, small groups keep CUs busy
        shell_ode_kernel<false><<<(rg.n + 63) / 64, 64, 0, stream>>>(m, rg, ShellGalaxy{}, ct, out);
    return hipGetLastError();
}

}  // namespace rt
