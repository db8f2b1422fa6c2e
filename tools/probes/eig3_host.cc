// Host build of leg-kilo_amd/csrc/lk_eig3.h for tests/test_eig3.py:  g++ -O2 -shared -fPIC -ffp-contract=off -o eig3_host.so eig3_host.cc
#include "../../leg-kilo_amd/csrc/lk_eig3.h"
extern "C" void lk_eig_sym3_host(const double* A6, double* ev3, double* V9) { lk_eig_sym3(A6, ev3, V9); }
extern "C" void lk_eig_sym3_host_n(const double* A6, double* ev3, double* V9, int n) {
    for (int i = 0; i < n; ++i) lk_eig_sym3(A6 + 6 * i, ev3 + 3 * i, V9 + 9 * i);
}
