// Host simulation of the wave-level FFT in aware_amd/csrc/fft512.hpp: the 64 lanes
// of a wavefront are run one after another between phase boundaries.  Built with
// hipcc (host code only) by tests/test_fft_host_sim.py; prints max errors against a
// naive double-precision DFT.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../aware_amd/csrc/fft512.hpp"

using namespace aware;

static std::vector<cf> tw512(512), tw1024(512);

// the LDS-table form used by the synthesis kernels
template <int DIR>
static void fft512_sim_tables(cf (*V)[8], cf* s) {
    std::vector<cf> t1(512), t2(64);
    for (int tid = 0; tid < 256; ++tid) fft_fill_tables(tid, 256, tw512.data(), t1.data(), t2.data());
    for (int l = 0; l < 64; ++l) fft_phaseA_t<DIR>(l, V[l], t1.data(), s);
    for (int l = 0; l < 64; ++l) fft_phaseB_t<DIR>(l, V[l], t2.data(), s);
    for (int l = 0; l < 64; ++l) fft_phaseB_store<DIR>(l, V[l], s);
    for (int l = 0; l < 64; ++l) fft_phaseC<DIR>(l, V[l], s);
}

template <int DIR>
static void fft512_sim(cf (*V)[8], cf* s) {
    FftLaneConst c[64];
    for (int l = 0; l < 64; ++l) fft_lane_const(l, tw512.data(), c[l]);
    for (int l = 0; l < 64; ++l) fft_phaseA<DIR>(l, V[l], c[l], s);
    for (int l = 0; l < 64; ++l) fft_phaseB<DIR>(l, V[l], c[l], s);
    for (int l = 0; l < 64; ++l) fft_phaseB_store<DIR>(l, V[l], s);
    for (int l = 0; l < 64; ++l) fft_phaseC<DIR>(l, V[l], s);
}

int main() {
    const double PI = 3.14159265358979323846;
    for (int j = 0; j < 512; ++j) {
        tw512[j] = mk((float)cos(2 * PI * j / 512), (float)-sin(2 * PI * j / 512));
        tw1024[j] = mk((float)cos(2 * PI * j / 1024), (float)-sin(2 * PI * j / 1024));
    }
    srand(7);
    std::vector<double> x(1024);
    for (auto& v : x) v = (rand() / (double)RAND_MAX) * 2 - 1;

    // ---------- forward: rfft1024 ----------
    static cf V[64][8];
    std::vector<cf> s(kFftScratch);
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 8; ++r) {
            int n = l + 64 * r;
            V[l][r] = mk((float)x[2 * n], (float)x[2 * n + 1]);
        }
    fft512_sim<-1>(V, s.data());
    for (int l = 0; l < 64; ++l) rfft_split_store(l, V[l], s.data());
    std::vector<cf> X(513);
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 8; ++r) {
            int k = l + 64 * r;
            X[k] = rfft_split_bin(k, V[l][r], s.data(), tw1024.data());
        }
    X[512] = mk(rfft_split_nyquist(s.data()), 0.f);
    double maxerr_f = 0, maxmag = 0;
    std::vector<double> Xr(513), Xi(513);
    for (int k = 0; k <= 512; ++k) {
        double re = 0, im = 0;
        for (int n = 0; n < 1024; ++n) {
            re += x[n] * cos(2 * PI * k * n / 1024);
            im -= x[n] * sin(2 * PI * k * n / 1024);
        }
        Xr[k] = re; Xi[k] = im;
        maxerr_f = fmax(maxerr_f, fmax(fabs(re - X[k].x), fabs(im - X[k].y)));
        maxmag = fmax(maxmag, hypot(re, im));
    }

    // ---------- inverse: irfft1024 of the exact spectrum ----------
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 8; ++r) {
            int k = l + 64 * r;
            cf xk = mk((float)Xr[k], (float)Xi[k]);
            cf xp = mk((float)Xr[512 - k], (float)Xi[512 - k]);
            if (k == 0) { xk.y = 0; xp.y = 0; }
            V[l][r] = irfft_merge_bin(k, xk, xp, tw1024.data());
        }
    fft512_sim_tables<1>(V, s.data());
    double maxerr_i = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 8; ++r) {
            int n = l + 64 * r;
            double a = V[l][r].x / 512.0, b = V[l][r].y / 512.0;
            maxerr_i = fmax(maxerr_i, fmax(fabs(a - x[2 * n]), fabs(b - x[2 * n + 1])));
        }
    printf("rfft_maxerr %.3e rfft_maxmag %.3e irfft_maxerr %.3e\n", maxerr_f, maxmag, maxerr_i);
    return 0;
}
