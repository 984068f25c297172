// rt_shell.hip -- the consumer of the Monte Carlo curve: the velocity of a radiation-driven gas shell,
// integrated for every H II region of a galaxy at once (one thread per region).
//
// Reference semantics (file:line into the reference checkout):
//   get_tau_IR          MC code.py:781-807    infrared optical depth of the shell
//   GetMC               MC code.py:843-863    momentum transfer function at (tau, log age): the curve runMC wrote
//   ODEs                MC code.py:905-955    dv/dr and dt/dr of the spherical shell
//   GetRegionVelocity   MC code.py:1433-1472  integration from h_g to 100 H_old, v(h_g) = v_0, t(h_g) = t_0
// The reference integrates one region at a time with scipy's Radau (max_step = 1 pc); here an embedded
// Dormand-Prince 5(4) pair with the same step cap runs one region per lane.  The system is not stiff;
// what makes it delicate is dv/dr ~ 1/v when a shell stalls, which step-size control handles.
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

// ODEs, MC code.py:913-955 (includeWholeGalaxy off)
__device__ void rhs(const ShellModel& m, const Region& q, double r, double v, double t, double& dvdr, double& dtdr) {
    const double new_time = log10(t);
    double l_bol = q.l_bol;
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
    dtdr = 1.0 / (v * m.year_s);
    if (r > m.r_escape) dvdr = -__builtin_inf();
}

__global__ void shell_rhs_kernel(const ShellModel m, const ShellRegions rg, int n, const int32_t* __restrict__ region,
                                 const double* __restrict__ r, const double* __restrict__ v,
                                 const double* __restrict__ t, double* __restrict__ dvdr, double* __restrict__ dtdr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Region q = load_region(rg, region[i]);
    double a, b;
    rhs(m, q, r[i], v[i], t[i], a, b);
    dvdr[i] = a;
    dtdr[i] = b;
}

// Dormand-Prince 5(4), one region per thread.  Steps are capped by max_step and land exactly on the
// n_out output radii r0 + (k+1)/n_out (r_end - r0).
__global__ void shell_ode_kernel(const ShellModel m, const ShellRegions rg, const ShellControl ct,
                                 ShellResult out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rg.n) return;
    const Region q = load_region(rg, i);
    // beyond r_escape the reference sets dv/dr = -inf and its solver stops (MC code.py:951-953): integrate to there
    const double r0 = q.h_g_i, r_end = rg.r_end[i], r_stop = fmin(r_end, m.r_escape);
    double r = r0, v = ct.v0, t = ct.t0;
    double v_max = v, r_vmax = r, t_vmax = t;
    int status = RT_SHELL_REACHED_END, steps = 0, next_out = 0;
    double k1v, k1t;
    rhs(m, q, r, v, t, k1v, k1t);
    double h = fmin(ct.max_step, 1e-3 * (r_end - r0));
    for (int k = 0; k < ct.n_out; ++k) {
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
        rhs(m, q, r + hs * (1.0 / 5.0), v + hs * (1.0 / 5.0) * k1v, t + hs * (1.0 / 5.0) * k1t, k2v, k2t);
        rhs(m, q, r + hs * (3.0 / 10.0), v + hs * (3.0 / 40.0 * k1v + 9.0 / 40.0 * k2v),
            t + hs * (3.0 / 40.0 * k1t + 9.0 / 40.0 * k2t), k3v, k3t);
        rhs(m, q, r + hs * (4.0 / 5.0), v + hs * (44.0 / 45.0 * k1v - 56.0 / 15.0 * k2v + 32.0 / 9.0 * k3v),
            t + hs * (44.0 / 45.0 * k1t - 56.0 / 15.0 * k2t + 32.0 / 9.0 * k3t), k4v, k4t);
        rhs(m, q, r + hs * (8.0 / 9.0),
            v + hs * (19372.0 / 6561.0 * k1v - 25360.0 / 2187.0 * k2v + 64448.0 / 6561.0 * k3v - 212.0 / 729.0 * k4v),
            t + hs * (19372.0 / 6561.0 * k1t - 25360.0 / 2187.0 * k2t + 64448.0 / 6561.0 * k3t - 212.0 / 729.0 * k4t),
            k5v, k5t);
        rhs(m, q, r + hs,
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
            rhs(m, q, r + hs, vn, tn, k7v, k7t);
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
                out.v_out[(size_t)i * ct.n_out + next_out] = v;
                out.t_out[(size_t)i * ct.n_out + next_out] = t;
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

hipError_t launch_shell_rhs(const ShellModel& m, const ShellRegions& rg, int n, const int32_t* d_region,
                            const double* d_r, const double* d_v, const double* d_t, double* d_dvdr, double* d_dtdr,
                            hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    shell_rhs_kernel<<<(n + 63) / 64, 64, 0, stream>>>(m, rg, n, d_region, d_r, d_v, d_t, d_dvdr, d_dtdr);
    return hipGetLastError();
}

hipError_t launch_shell_ode(const ShellModel& m, const ShellRegions& rg, const ShellControl& ct, const ShellResult& out,
                            hipStream_t stream) {
    if (rg.n <= 0) return hipSuccess;
    // 64 regions per workgroup: regions take very different numbers of steps, small groups keep CUs busy
    shell_ode_kernel<<<(rg.n + 63) / 64, 64, 0, stream>>>(m, rg, ct, out);
    return hipGetLastError();
}

}  // namespace rt
