/* encode_conventions.h -- THE frozen conventions of the permutohedral hash encoding, in one place.
 *
 * The arithmetic of the encoding lives in the un-vendored package github.com/RaduAlexandru/permutohedral_encoding (no
 * version pin; call sites permuto_sdf_py/models/models.py:20,149,154,186,333,442) whose source is absent from the
 * reference checkout: PARITY UNPINNED.  What the call sites do not force is frozen HERE and only
 * here.  This file is read by
 *   - the HIP kernels (encode_device.h, encode.hip, fused.hip: #include),
 *   - the host mirror (permuto_sdf_amd/conventions.py parses the #defines: scale_factor formula, parameter init, default
 *     concatenation layout),
 *   - the CPU oracle (oracle/permuto_oracle.py parses the same #defines),
 * so that diffing against the upstream source, the day it is available, is a change of THIS file: flip a value, rebuild,
 * and kernels, host code and oracle follow together (tests/test_encoding_conventions.py).
 *
 * Every line below is `#define NAME value  // explanation` with a plain numeric value (the parsers are that simple). */
#ifndef PSDF_ENCODE_CONVENTIONS_H
#define PSDF_ENCODE_CONVENTIONS_H

/* hash of a lattice vertex key: h = 0; for i < P: h += (uint32) key[i]; h *= MULTIPLIER; row = h % capacity   (App. A.3) */
#define PSDF_ENC_HASH_MULTIPLIER 2531011

/* rank sort of the residuals d_i = E_i - rem0_i, for i < j:  1: (d_i < d_j) ? rank[i]++ : rank[j]++  (a tie raises rank[j])
 *                                                            0: (d_i <= d_j) ? rank[i]++ : rank[j]++ (a tie raises rank[i]) */
#define PSDF_ENC_RANK_TIE_RAISES_LATER 1

/* scale_factor[l][i] = 1 / (TERM_i * scale_list[l]):  1: TERM_i = sqrt((i+1)(i+2))   0: TERM_i = 1          (App. A.2) */
#define PSDF_ENC_SCALE_SQRT_TERM 1
/* 1: additionally multiply scale_factor by the classic inverse standard deviation (P+1)*sqrt(2/3) of Adams et al. 2010 */
#define PSDF_ENC_SCALE_INV_STDDEV 0

/* layout of `concat_points=True` outputs for P position dims and F features per level:
 *   1: ceil(P/F) zero-padded pseudo-levels after the hashed ones: F*(L + ceil(P/F)) channels (52 for L=24, P=3, F=2)
 *   2: exactly P channels appended, cat([sliced, scaling * points]):  F*L + P channels (51)            (App. A.3 `final`)
 * Both layouts are built and tested; this is the DEFAULT the Python module uses when `concat_points=True`. */
#define PSDF_ENC_CONCAT_DEFAULT_LAYOUT 1

/* parameter initialisation: lattice_values = randn(T, L, F) * INIT permuted to [L, T, F]; shifts = randn(L, P) * SHIFT */
#define PSDF_ENC_LATTICE_INIT_SCALE 1e-5
#define PSDF_ENC_RANDOM_SHIFT_SCALE 10.0

/* values of the `concat_points` argument of the C ABI (include/psdf.h) */
#define PSDF_ENC_CONCAT_NONE 0
#define PSDF_ENC_CONCAT_PSEUDO_LEVELS 1
#define PSDF_ENC_CONCAT_APPEND 2

#endif
