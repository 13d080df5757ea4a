// CPU emulation of the FftFilter tile transform (rustradio_amd/csrc/fft_core.hpp):
// runs the exact per-thread pass functions the HIP kernel runs, thread by thread,
// and checks forward (digit-reversed) and inverse (natural) results against a
// naive f64 DFT.  Build: g++ -O2 -std=c++17 -I rustradio_amd/csrc tests/cpp/emu_fft.cpp
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_core.hpp"

using namespace rr;

template <int LOG2F, int I> void phase_fwd(std::vector<cf>& lds, const std::vector<cf>& x, const std::vector<cf>& tw,
                                           const std::vector<cf>& hpos, bool apply) {
    using G = PassGeom<LOG2F, I>;
    constexpr int NP = Plan<LOG2F>::NP;
    for (int t = 0; t < G::T; t++) {
        cf v[16];
        if (I == 0) { for (int n = 0; n < 16; n++) v[n] = x[n * G::T + t]; }
        else lds_load<LOG2F, I>(v, t, lds.data());
        cf twl[15], h[16];
        load_twiddles<LOG2F, I>(twl, t, tw.data());
        fwd_pass<LOG2F, I>(v, twl);
        if (I == NP - 1 && apply) { load_h<LOG2F, I>(h, t, hpos.data()); apply_h(v, h); }
        lds_store<LOG2F, I>(v, t, lds.data());
    }
}
template <int LOG2F, int I> void phase_inv(std::vector<cf>& lds, std::vector<cf>& y, const std::vector<cf>& tw) {
    using G = PassGeom<LOG2F, I>;
    for (int t = 0; t < G::T; t++) {
        cf v[16];
        lds_load<LOG2F, I>(v, t, lds.data());
        cf twl[15];
        load_twiddles<LOG2F, I>(twl, t, tw.data());
        inv_pass<LOG2F, I>(v, twl);
        if (I == 0) { for (int n = 0; n < 16; n++) y[n * G::T + t] = v[n]; }
        else lds_store<LOG2F, I>(v, t, lds.data());
    }
}

template <int LOG2F, int I> int check_addr() {
    using G = PassGeom<LOG2F, I>;
    int bad = 0;
    for (int t = 0; t < G::T; t++)
        for (int u = 0; u < G::U; u++)
            for (int n = 0; n < G::R; n++)
                if (lds_pad(G::pos(t + G::T * u, n)) != lds_base<LOG2F, I>(t) + lds_off<LOG2F, I>(u, n)) bad++;
    if (bad) printf("F=%d pass %d: %d LDS address mismatches\n", 1 << LOG2F, I, bad);
    return bad ? 1 : 0;
}

template <int LOG2F> int run() {
    constexpr int F = 1 << LOG2F;
    constexpr int NP = Plan<LOG2F>::NP;
    std::vector<cf> x(F), tw(F), lds(lds_elems(F)), y(F), hpos(F);
    std::vector<std::complex<double>> xd(F), H(F);
    srand(LOG2F);
    for (int i = 0; i < F; i++) {
        x[i] = mk((float)rand() / RAND_MAX * 2 - 1, (float)rand() / RAND_MAX * 2 - 1);
        xd[i] = {x[i].x, x[i].y};
        double a = -2.0 * M_PI * i / F;
        tw[i] = mk((float)cos(a), (float)sin(a));
        H[i] = {(double)rand() / RAND_MAX - 0.5, (double)rand() / RAND_MAX - 0.5};
    }
    for (int p = 0; p < F; p++) { auto h = H[bin_of_pos<LOG2F>(p)]; hpos[p] = mk((float)h.real(), (float)h.imag()); }
    // naive DFT (f64)
    std::vector<std::complex<double>> X(F), Y(F);
    for (int k = 0; k < F; k++) {
        std::complex<double> s = 0;
        for (int n = 0; n < F; n++) s += xd[n] * std::polar(1.0, -2.0 * M_PI * (double)((long)k * n % F) / F);
        X[k] = s;
    }
    for (int n = 0; n < F; n++) {
        std::complex<double> s = 0;
        for (int k = 0; k < F; k++) s += X[k] * H[k] * std::polar(1.0, 2.0 * M_PI * (double)((long)k * n % F) / F);
        Y[n] = s;
    }
    int fails = 0;
    fails += check_addr<LOG2F, 0>() + check_addr<LOG2F, 1>() + check_addr<LOG2F, 2>();
    if constexpr (NP > 3) fails += check_addr<LOG2F, 3>();
    for (int apply = 0; apply < 2; apply++) {
        phase_fwd<LOG2F, 0>(lds, x, tw, hpos, apply);
        phase_fwd<LOG2F, 1>(lds, x, tw, hpos, apply);
        phase_fwd<LOG2F, 2>(lds, x, tw, hpos, apply);
        if constexpr (NP > 3) phase_fwd<LOG2F, 3>(lds, x, tw, hpos, apply);
        if (!apply) {
            double emax = 0, xmax = 0;
            for (int p = 0; p < F; p++) {
                cf g = lds[lds_pad(p)];
                auto r = X[bin_of_pos<LOG2F>(p)];
                emax = fmax(emax, std::abs(std::complex<double>(g.x, g.y) - r));
                xmax = fmax(xmax, std::abs(r));
            }
            printf("F=%d forward  max-norm err %.3g\n", F, emax / xmax);
            if (!(emax / xmax < 2e-6)) fails++;
        } else {
            if constexpr (NP > 3) phase_inv<LOG2F, 3>(lds, y, tw);
            phase_inv<LOG2F, 2>(lds, y, tw);
            phase_inv<LOG2F, 1>(lds, y, tw);
            phase_inv<LOG2F, 0>(lds, y, tw);
            double emax = 0, ymax = 0;
            for (int n = 0; n < F; n++) {
                emax = fmax(emax, std::abs(std::complex<double>(y[n].x, y[n].y) - Y[n]));
                ymax = fmax(ymax, std::abs(Y[n]));
            }
            printf("F=%d fwd*H*inv max-norm err %.3g\n", F, emax / ymax);
            if (!(emax / ymax < 3e-6)) fails++;
        }
    }
    return fails;
}

int main() {
    int f = 0;
    // small-DFT unit checks
    for (int R : {2, 4, 8, 16}) {
        for (int inv = 0; inv < 2; inv++) {
            cf v[16]; std::complex<double> d[16];
            for (int i = 0; i < R; i++) { v[i] = mk((float)(i * 0.37 - 1), (float)(0.11 * i * i - 0.5)); d[i] = {v[i].x, v[i].y}; }
            if (R == 2) inv ? Dft<2, true>::run(v) : Dft<2, false>::run(v);
            if (R == 4) inv ? Dft<4, true>::run(v) : Dft<4, false>::run(v);
            if (R == 8) inv ? Dft<8, true>::run(v) : Dft<8, false>::run(v);
            if (R == 16) inv ? Dft<16, true>::run(v) : Dft<16, false>::run(v);
            double e = 0;
            for (int k = 0; k < R; k++) {
                std::complex<double> s = 0;
                for (int n = 0; n < R; n++) s += d[n] * std::polar(1.0, (inv ? 2.0 : -2.0) * M_PI * k * n / R);
                e = fmax(e, std::abs(s - std::complex<double>(v[k].x, v[k].y)));
            }
            printf("dft%d inv=%d err %.3g\n", R, inv, e);
            if (!(e < 1e-5)) f++;
        }
    }
    f += run<10>(); f += run<11>(); f += run<12>(); f += run<13>(); f += run<14>();
    printf(f ? "FAIL %d\n" : "OK\n", f);
    return f ? 1 : 0;
}
