// Test driver (tests/test_pnp.py::test_native_library_is_clean_under_asan_ubsan): 200 random 9-point pose problems
// through pvnet_pnp_solve, built together with pvnet_amd/csrc/pvnet_pnp.cpp under -fsanitize=address,undefined.
#include "pvnet_pnp.h"
#include <cstdio>
#include <cmath>
int main() {
    double K[9] = {572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1};
    double X[27], x[18], out[6];
    unsigned s = 1;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
    int fails = 0;
    for (int trial = 0; trial < 200; ++trial) {
        double aa[3] = {rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1}, R[9], t[3] = {rnd() * .4 - .2, rnd() * .4 - .2, .6 + rnd()};
        pvnet_angle_axis_to_matrix(aa, R);
        for (int i = 0; i < 9; ++i) {
            for (int a = 0; a < 3; ++a) X[3 * i + a] = rnd() * .16 - .08;
            double Y[3];
            for (int a = 0; a < 3; ++a) Y[a] = R[3 * a] * X[3 * i] + R[3 * a + 1] * X[3 * i + 1] + R[3 * a + 2] * X[3 * i + 2] + t[a];
            x[2 * i] = K[0] * Y[0] / Y[2] + K[2] + (rnd() - .5);
            x[2 * i + 1] = K[4] * Y[1] / Y[2] + K[5] + (rnd() - .5);
        }
        if (pvnet_pnp_solve(x, X, nullptr, K, out, 9) < 0) ++fails;
        double R2[9];
        pvnet_angle_axis_to_matrix(out, R2);
        double e = 0;
        for (int a = 0; a < 9; ++a) e = fmax(e, fabs(R2[a] - R[a]));
        if (e > 0.2 || fabs(out[5] - t[2]) > 0.2) ++fails;  // (a noisy small distant object can land in a local minimum)
    }
    printf("sanitized run done, %d of 200 poses off\n", fails);
    return fails > 4;
}
