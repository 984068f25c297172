// A caller of librt_amd's C ABI without Python: every entry point of include/rt_amd.h once or more, with the
// checks a caller can make on its own (conservation of photon histories, static sweep = allocating sweep bit for
// bit, reruns identical, error codes).  Test infrastructure: tests/test_gpu_native_driver.py runs the binary on the
// GPU box; built with host sanitizers (tools/asan_host.sh) it is the memory check of the library's host side.
// Exit code 0 = all checks passed; each failure prints a line.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "rt_amd.h"

static int g_failures = 0;
#define CHECK(cond, ...)                         \
    do {                                         \
        if (!(cond)) {                           \
            ++g_failures;                        \
            std::printf("FAIL %s:%d  ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);            \
            std::printf("\n");                   \
        }                                        \
    } while (0)
#define OK(call) CHECK((call) == RT_OK, "%s -> %s", #call, rt_last_error())
#define HIP(call) CHECK((call) == hipSuccess, "%s failed", #call)

struct Sweep {
    std::vector<double> tau;
    std::vector<uint64_t> n;
    rt_params p;
    Sweep(std::vector<double> t, uint64_t photons) : tau(std::move(t)), n(tau.size(), photons) {
        std::memset(&p, 0, sizeof p);
        p.struct_size = sizeof p;
        p.n_atm = (int32_t)tau.size();
        p.n_mu_bins = 10;
        p.n_slabs = 8;
        p.world = 1;
        p.seed = 0x5EED2026ull;
        p.alpha = 0.3;
        p.h = 6.6261e-27;
        p.tau_atm = tau.data();
        p.n_photons = n.data();
    }
};

struct Result {
    std::vector<double> sums, slab, dmom, den;
    std::vector<uint64_t> counts, hist;
    rt_tallies t;
    explicit Result(const rt_params& p) {
        const size_t a = (size_t)p.n_atm, ns = (size_t)(p.n_slabs > 0 ? p.n_slabs : 8);
        sums.assign(a * RT_NSUM, -1.0);
        slab.assign(a * ns * RT_NSUM, -1.0);
        counts.assign(a * RT_NCNT, 99);
        hist.assign(a * ((size_t)p.n_mu_bins + 1), 99);
        dmom.assign(a * (size_t)p.n_depth_bins + 1, -1.0);
        den.assign(a * (size_t)p.n_depth_bins + 1, -1.0);
        t = {sums.data(), counts.data(), hist.data(), p.n_depth_bins ? dmom.data() : nullptr,
             p.n_depth_bins ? den.data() : nullptr, slab.data(), nullptr};
    }
    bool histories_end_once(int n_atm) const {
        for (int a = 0; a < n_atm; ++a) {
            const uint64_t* c = &counts[(size_t)a * RT_NCNT];
            if (c[RT_CNT_ESCAPED] + c[RT_CNT_ESCAPED_UNCOUNTED] + c[RT_CNT_ABSORBED] + c[RT_CNT_GUARD] +
                    c[RT_CNT_BOUNDARY_ABSORBED] != c[RT_CNT_PHOTONS])
                return false;
        }
        return true;
    }
};

static void fill_tables(int n_tables, int n_wl, std::vector<double>& wl, std::vector<double>& g, std::vector<double>& alb,
                        std::vector<double>& sc, std::vector<double>& pdf) {
    wl.resize(n_wl);
    g.resize((size_t)n_tables * n_wl);
    alb.resize(g.size());
    sc.resize(g.size());
    pdf.resize(g.size());
    for (int i = 0; i < n_wl; ++i) wl[i] = std::round((0.1 + 0.9 * i / (n_wl - 1.0)) * 1e4) / 1e4;
    for (int t = 0; t < n_tables; ++t)
        for (int i = 0; i < n_wl; ++i) {
            const size_t k = (size_t)t * n_wl + i;
            const double x = i / (n_wl - 1.0);
            g[k] = (0.3 + 0.5 * x) / (1 + t);
            alb[k] = 0.4 + 0.5 * x;
            sc[k] = 2.0 - 1.5 * x + 0.1 * t;
            pdf[k] = (1.0 - 0.9 * x) / n_wl;
        }
}

int main() {
    CHECK((rt_abi_version() & 0xfff) == RT_ABI_VERSION, "ABI %d vs header %d", rt_abi_version(), RT_ABI_VERSION);
    rt_ctx* ctx = nullptr;
    if (rt_create(&ctx, 0) != RT_OK) {
        std::printf("rt_create: %s\n", rt_last_error());
        return 2;
    }
    char name[256];
    int n_cu = 0, wave = 0, khz = 0;
    OK(rt_device_info(ctx, name, sizeof name, &n_cu, &wave, &khz));
    std::printf("device %s, %d CUs, wave %d\n", name, n_cu, wave);
    CHECK(wave == 64 && n_cu > 0, "device info");

    // ---- call order and argument errors: refused with a message, nothing crashes ----
    {
        Sweep s({1.0}, 100);
        Result r(s.p);
        CHECK(rt_run(ctx, &s.p, &r.t) == RT_ERR_STATE, "run before tables must be RT_ERR_STATE");
        CHECK(std::strlen(rt_last_error()) > 0, "no message");
        CHECK(rt_run(ctx, nullptr, &r.t) == RT_ERR_INVALID, "NULL params");
        {   // a struct typed from an older header: its first word is n_atm, not a size -- refused before anything else is read
            Sweep old({1.0}, 100);
            old.p.struct_size = 1;
            CHECK(rt_run(ctx, &old.p, &r.t) == RT_ERR_INVALID && std::strstr(rt_last_error(), "struct_size"), "stale rt_params must be refused");
        }
        CHECK(rt_run(nullptr, &s.p, &r.t) == RT_ERR_INVALID, "NULL ctx");
    }
    std::vector<double> wl, g, alb, sc, pdf;
    fill_tables(1, 64, wl, g, alb, sc, pdf);
    OK(rt_set_tables(ctx, 64, wl.data(), g.data(), alb.data(), sc.data(), pdf.data()));
    {
        Sweep s({1.0}, 100);
        Result r(s.p);
        s.p.boundary = 7;
        CHECK(rt_run(ctx, &s.p, &r.t) == RT_ERR_INVALID, "bad boundary");
        s.p.boundary = 0;
        s.p.n_atm = 0;
        CHECK(rt_run(ctx, &s.p, &r.t) == RT_ERR_INVALID, "n_atm 0");
        s.p.n_atm = 1;
        s.p.n_mu_bins = 5000;
        CHECK(rt_run(ctx, &s.p, &r.t) == RT_ERR_LIMIT, "n_mu_bins limit");
        s.p.n_mu_bins = 10;
        s.p.world = 3;
        CHECK(rt_run(ctx, &s.p, &r.t) == RT_ERR_INVALID, "n_slabs %% world");
    }

    // ---- the sweep, every boundary x law x order, with and without depth bins; reruns identical ----
    for (int order = 0; order < 2; ++order)
        for (int boundary = 0; boundary < 3; ++boundary)
            for (int law = 0; law < 2; ++law)
                for (int nd = 0; nd <= 8; nd += 8) {
                    Sweep s({1e-3, 0.4, 3.0, 40.0}, 5000);
                    s.p.loop_order = order;
                    s.p.boundary = boundary;
                    s.p.scatter = law;
                    s.p.n_depth_bins = nd;
                    s.p.rand_mu = boundary == RT_BOUNDARY_REEMIT;
                    Result a(s.p), b(s.p);
                    OK(rt_run(ctx, &s.p, &a.t));
                    OK(rt_run(ctx, &s.p, &b.t));
                    CHECK(a.histories_end_once(s.p.n_atm), "histories order %d boundary %d law %d", order, boundary, law);
                    CHECK(a.counts[RT_CNT_PHOTONS] == 5000, "photon count");
                    CHECK(a.sums == b.sums && a.counts == b.counts && a.hist == b.hist && a.slab == b.slab, "rerun differs");
                    const char* k = rt_last_kernel(ctx);
                    CHECK(k && std::strstr(k, "walk_kernel"), "last kernel '%s'", k ? k : "(null)");
                }

    // ---- several tables, the other stream, per-photon escape cosines, many atmospheres ----
    {
        fill_tables(3, 64, wl, g, alb, sc, pdf);
        OK(rt_set_tables_multi(ctx, 3, 64, wl.data(), g.data(), alb.data(), sc.data(), pdf.data()));
        Sweep s({0.1, 1.0, 10.0, 1.0, 0.1}, 4000);
        const int32_t which[5] = {0, 1, 2, 2, 1};
        s.p.table_of_atm = which;
        Result r(s.p);
        OK(rt_run(ctx, &s.p, &r.t));
        CHECK(r.histories_end_once(5), "multi-table histories");
        const int32_t bad[5] = {0, 1, 3, 2, 1};
        s.p.table_of_atm = bad;
        CHECK(rt_run(ctx, &s.p, &r.t) == RT_ERR_INVALID, "table index out of range");
        s.p.table_of_atm = which;
        s.p.rng_mode = RT_RNG_PHILOX32X4;
        OK(rt_run(ctx, &s.p, &r.t));
        CHECK(r.histories_end_once(5), "packed stream histories");
        s.p.rng_mode = RT_RNG_PHILOX52;
        std::vector<double> emu(5 * 4000, 0.0);
        s.p.escaped_mu = emu.data();
        OK(rt_run(ctx, &s.p, &r.t));
        size_t logged = 0;
        for (double v : emu) logged += !std::isnan(v);
        uint64_t second = 0;
        for (int a = 0; a < 5; ++a) second += r.counts[(size_t)a * RT_NCNT + RT_CNT_ESCAPED_UNCOUNTED];
        CHECK(logged == second, "escaped_mu entries %zu vs second-test escapes %llu", logged, (unsigned long long)second);
        s.p.escaped_mu = nullptr;

        fill_tables(1, 64, wl, g, alb, sc, pdf);
        OK(rt_set_tables(ctx, 64, wl.data(), g.data(), alb.data(), sc.data(), pdf.data()));
        std::vector<double> many(5000);
        for (size_t i = 0; i < many.size(); ++i) many[i] = std::pow(10.0, -3.0 + 5.0 * i / (many.size() - 1.0));
        Sweep m(many, 50);
        m.n[17] = 0;
        m.n[4999] = 7777;
        Result rm(m.p);
        OK(rt_run(ctx, &m.p, &rm.t));
        CHECK(rm.histories_end_once(5000), "many atmospheres");
        CHECK(rm.counts[17 * RT_NCNT + RT_CNT_PHOTONS] == 0 && rm.counts[4999 * RT_NCNT + RT_CNT_PHOTONS] == 7777, "ragged counts");
    }

    // ---- results left on the device: allocating and static entry points agree bit for bit; ranks add up ----
    {
        Sweep s({0.01, 0.5, 5.0}, 60000);
        const size_t a = 3, n_slab = a * 8 * RT_NSUM, n_cnt = a * RT_NCNT, n_hist = a * 11;
        const size_t words = n_slab + n_cnt + n_hist;
        uint64_t *d1 = nullptr, *d2 = nullptr;
        CHECK(hipMalloc((void**)&d1, 8 * words) == hipSuccess && hipMalloc((void**)&d2, 8 * words) == hipSuccess, "hipMalloc");
        std::vector<uint64_t> h1(words), h2(words), acc(words, 0);
        OK(rt_run_device(ctx, &s.p, d1, d1 + n_slab, d1 + n_slab + n_cnt, nullptr, nullptr));
        OK(rt_run_device_static(ctx, &s.p, d2, d2 + n_slab, d2 + n_slab + n_cnt, nullptr, nullptr));
        HIP(hipDeviceSynchronize());
        HIP(hipMemcpy(h1.data(), d1, 8 * words, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(h2.data(), d2, 8 * words, hipMemcpyDeviceToHost));
        CHECK(h1 == h2, "static sweep differs from the allocating one");
        for (int rank = 0; rank < 4; ++rank) {  // four ranks' shares: summed as integers they give the one-rank buffer
            s.p.world = 4;
            s.p.rank = rank;
            OK(rt_run_device_static(ctx, &s.p, d2, d2 + n_slab, d2 + n_slab + n_cnt, nullptr, nullptr));
            HIP(hipDeviceSynchronize());
            HIP(hipMemcpy(h2.data(), d2, 8 * words, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < words; ++i) acc[i] += h2[i];
        }
        CHECK(acc == h1, "rank shares do not add up to the whole");
        s.p.world = 1;
        s.p.rank = 0;
        std::vector<double> t5000(5000, 1.0);
        Sweep big(t5000, 10);
        CHECK(rt_run_device_static(ctx, &big.p, d1, d1, d1, nullptr, nullptr) == RT_ERR_LIMIT, "static n_atm limit");

        // the library's own collective with a communicator of one
        unsigned char id[RT_COMM_ID_BYTES];
        if (rt_comm_unique_id(id) == RT_OK) {
            OK(rt_comm_init(ctx, 0, 1, id));
            OK(rt_run_allreduce(ctx, &s.p, d2, nullptr));
            HIP(hipDeviceSynchronize());
            HIP(hipMemcpy(h2.data(), d2, 8 * words, hipMemcpyDeviceToHost));
            CHECK(h2 == h1, "all-reduce over one rank changed the tallies");
            OK(rt_comm_destroy(ctx));
        } else {
            std::printf("note: no RCCL here (%s); collective not exercised\n", rt_last_error());
        }
        HIP(hipFree(d1));
        HIP(hipFree(d2));
    }

    // ---- replay on given uniforms ----
    {
        const int64_t n = 300;
        const int32_t max_draws = 48;
        std::mt19937_64 gen(5);
        std::uniform_real_distribution<double> u(0.0, 1.0);
        std::vector<double> xi((size_t)n * max_draws), lam(n, 0.5), gg(n, 0.6), al(n, 0.8), scl(n, 1.0), mu0(n, 1.0);
        for (double& x : xi) x = u(gen);
        Sweep s({2.0}, (uint64_t)n);
        Result r(s.p);
        std::vector<rt_photon_trace> tr(n);
        OK(rt_replay(ctx, &s.p, n, max_draws, xi.data(), lam.data(), gg.data(), al.data(), scl.data(), mu0.data(), &r.t, tr.data()));
        uint64_t ended = 0;
        for (const auto& t : tr) ended += t.fate != 0 && t.n_draws <= max_draws;
        CHECK(ended == (uint64_t)n && r.counts[RT_CNT_PHOTONS] == (uint64_t)n, "replay: %llu of %lld photons ended", (unsigned long long)ended, (long long)n);
    }

    // ---- table builder ----
    {
        const double a[3] = {0.01, 0.1, 1.0};
        const int32_t off[4] = {0, 4, 8, 12};
        std::vector<double> w, qa, qs, gq;
        for (int sz = 0; sz < 3; ++sz)
            for (int k = 0; k < 4; ++k) {
                w.push_back(10.0 / std::pow(10.0, k));  // descending: 10, 1, 0.1, 0.01
                qa.push_back(0.1 * (sz + 1) * (k + 1));
                qs.push_back(0.2 * (sz + 1) * (k + 1));
                gq.push_back(0.1 * k + 0.05 * sz);
            }
        const rt_grain_set gs = {3, a, off, w.data(), qa.data(), qs.data(), gq.data()};
        std::vector<double> grid(50), out[6];
        for (int i = 0; i < 50; ++i) grid[i] = 0.02 + 0.1 * i;
        for (auto& o : out) o.assign(50, -1.0);
        rt_grain_tables gt = {out[0].data(), out[1].data(), out[2].data(), out[3].data(), out[4].data(), out[5].data(), 0, 0, 0};
        OK(rt_build_grain_tables(ctx, &gs, 3.3e-4, 0.001, 1.0, 1, 50, grid.data(), nullptr, &gt));
        bool fin = true;
        for (auto& o : out)
            for (double v : o) fin = fin && std::isfinite(v) && v >= 0.0;
        CHECK(fin && gt.lum > 0.0, "grain tables");
    }

    // ---- shell velocities: alone, and with the galaxy's other regions ----
    {
        const double mc_tau[4] = {1e-3, 0.1, 1.0, 100.0}, mc_age[2] = {6.0, 7.0};
        const double mc_val[8] = {1e-3, 2e-3, 0.09, 0.1, 0.6, 0.7, 1.2, 1.4};
        const double l_age[3] = {6.0, 6.5, 7.0}, l_val[3] = {1.0, 0.8, 0.3}, ratio[3] = {1.0, 1.1, 1.2};
        const double cm_pc = 3.086e18;
        rt_shell_model m;
        std::memset(&m, 0, sizeof m);
        m.struct_size = sizeof m;
        m.n_tau = 4; m.n_age = 2; m.n_lage = 3;
        m.G = 6.67259e-8; m.c = 2.99792458e10; m.sigma_sb = 5.67037441918443e-5; m.year_s = 31556952.0;
        m.h_hi_cm = 100 * cm_pc; m.r_escape = 10000 * cm_pc; m.time_slice = 6.0; m.l_norm = 1.0;
        m.mc_tau = mc_tau; m.mc_age = mc_age; m.mc_val = mc_val; m.l_age = l_age; m.l_val = l_val; m.ratio_val = ratio;
        const int n = 5;
        std::vector<double> m_g(n, 2e36), m_new(n, 8e35), m_old(n, 1e35), m_co(n, 0.0), h_old(n, 3e20), h_g(n, 5 * cm_pc),
            tau_i(n), l_bol(n), sig(n, 8e-4), r_end(n, 45 * cm_pc);
        for (int i = 0; i < n; ++i) {
            tau_i[i] = 0.05 * (i + 1);
            l_bol[i] = 1e40 * (i + 1);
        }
        const rt_shell_regions rg = {n, 0, m_g.data(), m_new.data(), m_old.data(), m_co.data(), h_old.data(), h_g.data(),
                                     tau_i.data(), l_bol.data(), sig.data(), r_end.data()};
        const rt_shell_control ct = {100.0, 1e6, 1e-8, 1e-6, 1e-6, cm_pc, 4, 200000};
        std::vector<double> o[6], vo(n * 4), to(n * 4);
        for (auto& x : o) x.assign(n, -1.0);
        std::vector<int32_t> st(n, -1), steps(n, -1);
        rt_shell_result res = {o[0].data(), o[1].data(), o[2].data(), o[3].data(), o[4].data(), o[5].data(), st.data(),
                               steps.data(), vo.data(), to.data()};
        for (int aging = 0; aging < 2; ++aging) {
            m.aging = aging;
            OK(rt_shell_velocity(ctx, &m, &rg, &ct, &res));
            for (int i = 0; i < n; ++i) CHECK(st[i] >= 0 && st[i] <= 3 && steps[i] > 0 && o[0][i] >= 100.0, "shell region %d", i);
        }
        const std::vector<double> alone = o[0];
        // the five regions as each other's neighbours
        std::vector<double> lha(n * n), dist(n * n), mass(n, 3e36);
        std::vector<int32_t> view(n), self(n);
        for (int i = 0; i < n; ++i) {
            view[i] = i;
            self[i] = i;
            for (int j = 0; j < n; ++j) {
                lha[i * n + j] = 0.5 * (j + 1.0) / (i + 1.0);
                dist[i * n + j] = 30.0 * std::abs(i - j);
            }
        }
        const rt_shell_galaxy gal = {n, n, cm_pc, 1.1, lha.data(), dist.data(), mass.data(), view.data(), self.data()};
        m.galaxy = &gal;
        OK(rt_shell_velocity(ctx, &m, &rg, &ct, &res));
        bool differs = false;
        for (int i = 0; i < n; ++i) differs = differs || o[0][i] != alone[i];
        CHECK(differs, "the galaxy terms changed nothing");
        const int32_t idx[3] = {0, 2, 4};
        const double rr[3] = {6 * cm_pc, 9 * cm_pc, 20 * cm_pc}, vv[3] = {1e3, 1e4, 1e5}, tt[3] = {1e6, 2e6, 5e6};
        double dv[3], dt[3];
        OK(rt_shell_rhs(ctx, &m, &rg, 3, idx, rr, vv, tt, dv, dt));
        CHECK(std::isfinite(dv[0]) && dt[1] > 0.0, "shell rhs");
        view[2] = 9;
        CHECK(rt_shell_rhs(ctx, &m, &rg, 3, idx, rr, vv, tt, dv, dt) == RT_ERR_INVALID, "bad view index");
    }

    // ---- the math self-test and the timing query ----
    {
        std::vector<double> uu(1000), l(1000), c(1000), sq(1000), dv(1000);
        for (int i = 0; i < 1000; ++i) uu[i] = (i + 0.5) / 1000.0;
        OK(rt_selftest_math(ctx, 1000, uu.data(), l.data(), c.data(), sq.data(), dv.data()));
        double worst = 0.0;
        for (int i = 0; i < 1000; ++i) worst = std::fmax(worst, std::fabs(l[i] - std::log(uu[i])) / std::fabs(std::log(uu[i])));
        CHECK(worst < 1e-13, "device ln off by %.3g", worst);  // relative, down to |ln u| = 5e-4
        double ms = 0.0;
        int64_t launches = 0;
        OK(rt_kernel_time(ctx, &ms, &launches, 1));
        CHECK(launches > 0 && ms > 0.0, "kernel time");
    }

    OK(rt_destroy(ctx));
    std::printf(g_failures ? "%d check(s) failed\n" : "all checks passed\n", g_failures);
    return g_failures ? 1 : 0;
}
