// lk_eig3.h — eigen-decomposition of a symmetric 3 x 3 matrix WITHOUT iteration (stand-in for Eigen::EigenSolver<Matrix3d> on the
// covariance of a voxel's points, voxel_map.cc:55; the reference's use is invariant to eigenvector sign and order).
//
// Why not Jacobi on the device: one plane fit is the serial work of ONE wave, and a cyclic Jacobi sweep is three rotations of
// 3 fp64 divisions + 2 square roots each - 6-7 sweeps to converge = ~100 dependent division / sqrt sequences (measured 3.8 us per
// fit inside lk_insert_apply_kernel).  The closed form below is one acos, two cos, and a handful of divisions / square roots:
//   eigenvalues   trigonometric solution of the characteristic cubic of the shifted, scaled matrix (B = (A - q I) / p has
//                 eigenvalues 2 cos(phi + 2 pi k / 3), det B / 2 = cos 3 phi);
//   eigenvectors  the best-conditioned one first (the eigenvalue farthest from the other two): largest cross product of two rows of
//                 A - lambda I; the second from the 2 x 2 problem in its orthogonal complement; the third as a cross product -
//                 so the basis is orthonormal to rounding even when two eigenvalues coincide.
// The eigenvalues are then replaced by the Rayleigh quotients of those vectors.  Absolute errors are a few ulp of the LARGEST
// eigenvalue (ev[] may leave ascending order by that much when two coincide); for the planes of a voxel map (lambda_min /
// lambda_max >= ~1e-4) that is ~1e-12 relative on lambda_min - far inside the 1e-6 the parity tests ask of plane parameters.  tests/test_eig3.py checks it
// against LAPACK on random, near-degenerate and rank-deficient inputs (compiled for the host by tools/probes/eig3_host.cc).
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define LK_HD __host__ __device__ __forceinline__
#else
#define LK_HD inline
#endif

// unit vector in the null space direction of the rank-2 matrix with rows r0, r1, r2 (symmetric: A - lambda I)
LK_HD void lk_eig3_vec0(double a00, double a01, double a02, double a11, double a12, double a22, double ev, double* v) {
    const double r0x = a00 - ev, r0y = a01, r0z = a02;
    const double r1x = a01, r1y = a11 - ev, r1z = a12;
    const double r2x = a02, r2y = a12, r2z = a22 - ev;
    const double c0x = r0y * r1z - r0z * r1y, c0y = r0z * r1x - r0x * r1z, c0z = r0x * r1y - r0y * r1x;   // r0 x r1
    const double c1x = r0y * r2z - r0z * r2y, c1y = r0z * r2x - r0x * r2z, c1z = r0x * r2y - r0y * r2x;   // r0 x r2
    const double c2x = r1y * r2z - r1z * r2y, c2y = r1z * r2x - r1x * r2z, c2z = r1x * r2y - r1y * r2x;   // r1 x r2
    const double d0 = c0x * c0x + c0y * c0y + c0z * c0z, d1 = c1x * c1x + c1y * c1y + c1z * c1z, d2 = c2x * c2x + c2y * c2y + c2z * c2z;
    double bx = c0x, by = c0y, bz = c0z, bd = d0;
    if (d1 > bd) bx = c1x, by = c1y, bz = c1z, bd = d1;
    if (d2 > bd) bx = c2x, by = c2y, bz = c2z, bd = d2;
    if (bd > 0.0) {
        const double s = 1.0 / sqrt(bd);
        v[0] = bx * s, v[1] = by * s, v[2] = bz * s;
    } else {   // A - ev I vanished (cannot happen for a matrix with non-zero off-diagonal part; kept total)
        v[0] = 1.0, v[1] = 0.0, v[2] = 0.0;
    }
}
// second eigenvector: the null direction of (A - ev1 I) restricted to the plane orthogonal to the unit vector w
LK_HD void lk_eig3_vec1(double a00, double a01, double a02, double a11, double a12, double a22, const double* w, double ev1, double* v) {
    // orthonormal basis u, t of the complement of w
    double ux, uy, uz;
    if (fabs(w[0]) > fabs(w[1])) {
        const double s = 1.0 / sqrt(w[0] * w[0] + w[2] * w[2]);
        ux = -w[2] * s, uy = 0.0, uz = w[0] * s;
    } else {
        const double s = 1.0 / sqrt(w[1] * w[1] + w[2] * w[2]);
        ux = 0.0, uy = w[2] * s, uz = -w[1] * s;
    }
    const double tx = w[1] * uz - w[2] * uy, ty = w[2] * ux - w[0] * uz, tz = w[0] * uy - w[1] * ux;   // w x u
    // A u, A t
    const double aux = a00 * ux + a01 * uy + a02 * uz, auy = a01 * ux + a11 * uy + a12 * uz, auz = a02 * ux + a12 * uy + a22 * uz;
    const double atx = a00 * tx + a01 * ty + a02 * tz, aty = a01 * tx + a11 * ty + a12 * tz, atz = a02 * tx + a12 * ty + a22 * tz;
    double m00 = ux * aux + uy * auy + uz * auz - ev1;
    double m01 = ux * atx + uy * aty + uz * atz;
    double m11 = tx * atx + ty * aty + tz * atz - ev1;
    // null vector (c0, c1) of the rank-1 2 x 2 matrix [[m00 m01] [m01 m11]]: orthogonal to its larger row
    const double am00 = fabs(m00), am01 = fabs(m01), am11 = fabs(m11);
    double c0, c1;
    if (am00 >= am11) {
        if (am00 > 0.0 || am01 > 0.0) {
            if (am00 >= am01) {
                const double r = m01 / m00;
                const double s = 1.0 / sqrt(1.0 + r * r);
                c0 = -r * s, c1 = s;       // (m00, m01) . (c0, c1) = 0
            } else {
                const double r = m00 / m01;
                const double s = 1.0 / sqrt(1.0 + r * r);
                c0 = s, c1 = -r * s;
            }
        } else {
            c0 = 1.0, c1 = 0.0;             // the restriction vanished: ev1 is a double eigenvalue, any direction serves
        }
    } else {
        if (am11 >= am01) {
            const double r = m01 / m11;
            const double s = 1.0 / sqrt(1.0 + r * r);
            c0 = s, c1 = -r * s;           // (m01, m11) . (c0, c1) = 0
        } else {
            const double r = m11 / m01;
            const double s = 1.0 / sqrt(1.0 + r * r);
            c0 = -r * s, c1 = s;
        }
    }
    v[0] = c0 * ux + c1 * tx, v[1] = c0 * uy + c1 * ty, v[2] = c0 * uz + c1 * tz;
}

// A = [a0 a1 a2; a1 a3 a4; a2 a4 a5] (upper triangle, row-major).  ev[3] ascending; v0, v1, v2 = unit eigenvectors of ev[0], ev[1], ev[2].
// (Three separate vectors, not a 3 x 3 array: on the device a caller that picks a column by a run-time index made the compiler keep
// the array in memory - promoted to LDS, 72 B per thread - and the map kernels ran 30-40 % longer for it.)
LK_HD void lk_eig_sym3_cols(const double* Ain, double* ev, double* v0, double* v1, double* v2) {
    double a00 = Ain[0], a01 = Ain[1], a02 = Ain[2], a11 = Ain[3], a12 = Ain[4], a22 = Ain[5];
    double mx = fmax(fmax(fabs(a00), fabs(a01)), fmax(fabs(a02), fabs(a11)));
    mx = fmax(mx, fmax(fabs(a12), fabs(a22)));
    if (!(mx > 0.0)) {   // zero matrix (or NaN input): identity basis
        ev[0] = ev[1] = ev[2] = (mx == 0.0) ? 0.0 : mx;
        v0[0] = 1, v0[1] = 0, v0[2] = 0, v1[0] = 0, v1[1] = 1, v1[2] = 0, v2[0] = 0, v2[1] = 0, v2[2] = 1;
        return;
    }
    const double inv = 1.0 / mx;
    a00 *= inv, a01 *= inv, a02 *= inv, a11 *= inv, a12 *= inv, a22 *= inv;
    const double nrm = a01 * a01 + a02 * a02 + a12 * a12;
    double e0, e1, e2;
    if (nrm > 0.0) {
        const double q = (a00 + a11 + a22) / 3.0;
        const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
        const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * nrm) / 6.0);
        const double c00 = b11 * b22 - a12 * a12, c01 = a01 * b22 - a12 * a02, c02 = a01 * a12 - b11 * a02;
        const double det = (b00 * c00 - a01 * c01 + a02 * c02) / (p * p * p);
        const double hd = fmin(fmax(0.5 * det, -1.0), 1.0);
        const double ang = acos(hd) / 3.0;
        const double twoThirdsPi = 2.09439510239319549;
        const double beta2 = 2.0 * cos(ang), beta0 = 2.0 * cos(ang + twoThirdsPi), beta1 = -(beta0 + beta2);
        e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;   // ascending
        // the isolated eigenvalue's vector first (e2 when hd >= 0, else e0), the middle one from its complement, the third as a cross
        // product: v0 = v1 x v2, or v2 = v0 x v1.  One copy of the arithmetic with the operands SELECTED (a branch per case writes
        // through a pointer that depends on the branch, which keeps the three vectors in memory on the device)
        const bool top = hd >= 0.0;
        double iso[3], oth[3];
        lk_eig3_vec0(a00, a01, a02, a11, a12, a22, top ? e2 : e0, iso);
        lk_eig3_vec1(a00, a01, a02, a11, a12, a22, iso, e1, v1);
        const double px = top ? v1[0] : iso[0], py = top ? v1[1] : iso[1], pz = top ? v1[2] : iso[2];   // p x q with (p, q) = (v1, v2) or (v0, v1)
        const double qx = top ? iso[0] : v1[0], qy = top ? iso[1] : v1[1], qz = top ? iso[2] : v1[2];
        oth[0] = py * qz - pz * qy, oth[1] = pz * qx - px * qz, oth[2] = px * qy - py * qx;
        for (int k = 0; k < 3; ++k) v2[k] = top ? iso[k] : oth[k], v0[k] = top ? oth[k] : iso[k];
    } else {   // diagonal already: sort the diagonal, unit vectors
        double d[3] = {a00, a11, a22};
        int i0 = 0, i1 = 1, i2 = 2;
        double d0 = d[0], d1 = d[1], d2 = d[2];   // sorted as scalars beside their indices (no run-time index into d[])
        if (d0 > d1) { int t = i0; i0 = i1; i1 = t; double u = d0; d0 = d1; d1 = u; }
        if (d1 > d2) { int t = i1; i1 = i2; i2 = t; double u = d1; d1 = d2; d2 = u; }
        if (d0 > d1) { int t = i0; i0 = i1; i1 = t; double u = d0; d0 = d1; d1 = u; }
        e0 = d0, e1 = d1, e2 = d2;
        for (int k = 0; k < 3; ++k) v0[k] = (k == i0) ? 1.0 : 0.0, v1[k] = (k == i1) ? 1.0 : 0.0, v2[k] = (k == i2) ? 1.0 : 0.0;
    }
    if (nrm > 0.0) {
        // Rayleigh quotients of the (orthonormal) eigenvectors: the trigonometric eigenvalues lose digits when two of them nearly
        // coincide (acos near +-1); v^T A v is second-order accurate in the eigenvector error
        auto rq = [&](const double* v) {
            const double ax = a00 * v[0] + a01 * v[1] + a02 * v[2], ay = a01 * v[0] + a11 * v[1] + a12 * v[2], az = a02 * v[0] + a12 * v[1] + a22 * v[2];
            return v[0] * ax + v[1] * ay + v[2] * az;
        };
        e0 = rq(v0), e1 = rq(v1), e2 = rq(v2);
    }
    ev[0] = e0 * mx, ev[1] = e1 * mx, ev[2] = e2 * mx;
}
// V row-major, COLUMN k = eigenvector of ev[k]
LK_HD void lk_eig_sym3(const double* Ain, double* ev, double* V) {
    double v0[3], v1[3], v2[3];
    lk_eig_sym3_cols(Ain, ev, v0, v1, v2);
    for (int k = 0; k < 3; ++k) V[3 * k + 0] = v0[k], V[3 * k + 1] = v1[k], V[3 * k + 2] = v2[k];
}
