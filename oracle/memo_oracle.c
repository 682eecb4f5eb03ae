/*
 * memo_oracle.c -- CPU restatement of MEMO's windowed k-mer query path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP product in
 * memo_amd/csrc/: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it.  Nothing under memo_amd/ links
 * or imports it, and the product never falls back to it.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle.py)
 * against the golden vectors in tests/golden/, which tools/make_golden.py made
 * by executing the reference's own src/memo_query.py (filter_pq, memo_init,
 * memo_query, print_res) in the authoring container.  The reference ships no
 * tests or known-answer vectors of its own.
 *
 * What is restated (citations are /root/reference/src/memo_query.py:line):
 *   oracle_filter          filter_pq                       :19-36  (+ call site :100)
 *   oracle_literal_*       memo_init + memo_query + argmax :42-55, :57-63, :70 / :68
 *                          -- same bool matrix, same loop nest, single thread:
 *                          this is "the reference CPU path" timed by bench.py
 *   oracle_closed_*        the closed forms (C)/(M) of SURVEY.md section 0
 *                          -- no L x (N+1) matrix; used where the literal form
 *                          cannot allocate it
 *   oracle_emit_*          print_res                       :65-71
 *   oracle_synth_rows      the synthetic pangenome index of DESIGN.md (not in
 *                          the reference; the same generator runs on the GPU)
 *
 * Result encodings (shared with include/memo_amd.h):
 *   conservation  uint16 out[L]          value = first set column, N if none
 *   membership    uint32 out[L*W], W = ceil(N/32); genome g of position p is
 *                 bit (g & 31) of word p*W + (g >> 5); bits >= N are 0
 *
 * Build:  make -C oracle      (gcc -O2, no dependencies)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_EINDEX (-1) /* the reference raises IndexError (NumPy) / is UB (Numba) */
#define ORACLE_ENOMEM (-2)
#define ORACLE_EVALUE (-3) /* qe < qs: np.zeros([negative, ..]) raises ValueError (:51/:53) */

static inline int64_t clip64(int64_t v, int64_t lo, int64_t hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}

/* ------------------------------------------------------------------------
 * filter_pq  (memo_query.py:19-36, called with query_end + k at :100).
 * Two predicates on (f1,f2) of rows of the query chromosome; output keeps the
 * reference's order: all first-arm rows, then all second-arm rows.
 *   arm 1  f1 <= qs  and  f2 > qs            (:22-24)
 *   arm 2  f1 >  qs  and  f1 < qe + k        (:25-27)
 * Returns the number of rows written to (fs,fe,fo), each sized >= m.
 * ---------------------------------------------------------------------- */
uint64_t oracle_filter(const int64_t *s, const int64_t *e, const int64_t *o, uint64_t m,
                       int64_t qs, int64_t qe, int64_t k,
                       int64_t *fs, int64_t *fe, int64_t *fo) {
    uint64_t n = 0;
    int64_t qend = qe + k;
    for (uint64_t i = 0; i < m; i++)
        if (s[i] <= qs && e[i] > qs) { fs[n] = s[i]; fe[n] = e[i]; fo[n] = o[i]; n++; }
    for (uint64_t i = 0; i < m; i++)
        if (s[i] > qs && s[i] < qend) { fs[n] = s[i]; fe[n] = e[i]; fo[n] = o[i]; n++; }
    return n;
}

/* ------------------------------------------------------------------------
 * Literal transcription.  rec is the reference's bool matrix:
 *   conservation  rec[L][N+1] zeros, column N preset to 1     (:53-54)
 *   membership    rec[L][N]   ones                            (:51)
 * Rows are recentred, shadow-cast and clipped (:46-48), rows with
 * casted_end >= start are dropped (:49), and each surviving row writes
 * rec[casted_end:start, order] (:61-62).  NumPy/Numba index semantics for the
 * column: a negative order wraps once (order + ncols); anything still outside
 * [0, ncols) is an IndexError.
 * ---------------------------------------------------------------------- */
static int literal_fill(const int64_t *s, const int64_t *e, const int64_t *o, uint64_t m,
                        int64_t qs, int64_t qe, int64_t k, int64_t ncols,
                        uint8_t *rec, uint8_t set_bit) {
    int64_t L = qe - qs;
    for (uint64_t i = 0; i < m; i++) {
        int64_t st = clip64(s[i] - qs, 0, L);
        int64_t ce = clip64(e[i] - qs - (k - 1), 0, L);
        if (!(ce < st)) continue;
        int64_t col = o[i];
        if (col < 0) col += ncols;
        if (col < 0 || col >= ncols) return ORACLE_EINDEX;
        for (int64_t p = ce; p < st; p++) rec[p * ncols + col] = set_bit;
    }
    return ORACLE_OK;
}

int oracle_literal_conservation(const int64_t *s, const int64_t *e, const int64_t *o, uint64_t m,
                                int64_t qs, int64_t qe, int64_t k, int64_t N, uint16_t *out) {
    int64_t L = qe - qs;
    if (L < 0) return ORACLE_EVALUE;
    if (L == 0) return ORACLE_OK;
    int64_t nc = N + 1;
    uint8_t *rec = (uint8_t *)calloc((size_t)L * (size_t)nc, 1);
    if (!rec) return ORACLE_ENOMEM;
    for (int64_t p = 0; p < L; p++) rec[p * nc + N] = 1;
    int rc = literal_fill(s, e, o, m, qs, qe, k, nc, rec, 1);
    if (rc == ORACLE_OK)
        for (int64_t p = 0; p < L; p++) { /* np.argmax(rec, axis=1)  (:70) */
            const uint8_t *row = rec + p * nc;
            int64_t c = 0;
            while (!row[c]) c++;
            out[p] = (uint16_t)c;
        }
    free(rec);
    return rc;
}

int oracle_literal_membership(const int64_t *s, const int64_t *e, const int64_t *o, uint64_t m,
                              int64_t qs, int64_t qe, int64_t k, int64_t N, uint32_t *out) {
    int64_t L = qe - qs;
    if (L < 0) return ORACLE_EVALUE;
    if (L == 0) return ORACLE_OK;
    int64_t W = (N + 31) / 32;
    uint8_t *rec = (uint8_t *)malloc((size_t)L * (size_t)N);
    if (!rec) return ORACLE_ENOMEM;
    memset(rec, 1, (size_t)L * (size_t)N);
    int rc = literal_fill(s, e, o, m, qs, qe, k, N, rec, 0);
    if (rc == ORACLE_OK) {
        memset(out, 0, (size_t)L * (size_t)W * sizeof(uint32_t));
        for (int64_t p = 0; p < L; p++)
            for (int64_t g = 0; g < N; g++)
                if (rec[p * N + g]) out[p * W + (g >> 5)] |= (uint32_t)1 << (g & 31);
    }
    free(rec);
    return rc;
}

/* ------------------------------------------------------------------------
 * Closed forms (eqs. C and M): no matrix, O(m*k + L).
 *   out[p] = min( {o_i : lo_i <= p < hi_i} U {N} )
 *   bit[p][g] = 0 iff some row with o_i == g covers p
 * with hi_i = clip(s_i-qs, 0, L), lo_i = clip(e_i-qs-(k-1), 0, L).
 * Applied to the rows as given (call oracle_filter first to mirror main()).
 * ---------------------------------------------------------------------- */
int oracle_closed_conservation(const int64_t *s, const int64_t *e, const int64_t *o, uint64_t m,
                               int64_t qs, int64_t qe, int64_t k, int64_t N, uint16_t *out) {
    int64_t L = qe - qs, nc = N + 1;
    if (L < 0) return ORACLE_EVALUE;
    for (int64_t p = 0; p < L; p++) out[p] = (uint16_t)N;
    for (uint64_t i = 0; i < m; i++) {
        int64_t hi = clip64(s[i] - qs, 0, L), lo = clip64(e[i] - qs - (k - 1), 0, L);
        if (!(lo < hi)) continue;
        int64_t col = o[i];
        if (col < 0) col += nc;
        if (col < 0 || col >= nc) return ORACLE_EINDEX;
        for (int64_t p = lo; p < hi; p++)
            if (col < out[p]) out[p] = (uint16_t)col;
    }
    return ORACLE_OK;
}

int oracle_closed_membership(const int64_t *s, const int64_t *e, const int64_t *o, uint64_t m,
                             int64_t qs, int64_t qe, int64_t k, int64_t N, uint32_t *out) {
    int64_t L = qe - qs, W = (N + 31) / 32;
    if (L < 0) return ORACLE_EVALUE;
    for (int64_t p = 0; p < L; p++)
        for (int64_t w = 0; w < W; w++) {
            int64_t left = N - 32 * w;
            out[p * W + w] = left >= 32 ? 0xFFFFFFFFu : (((uint32_t)1 << left) - 1u);
        }
    for (uint64_t i = 0; i < m; i++) {
        int64_t hi = clip64(s[i] - qs, 0, L), lo = clip64(e[i] - qs - (k - 1), 0, L);
        if (!(lo < hi)) continue;
        int64_t col = o[i];
        if (col < 0) col += N;
        if (col < 0 || col >= N) return ORACLE_EINDEX;
        uint32_t msk = ~((uint32_t)1 << (col & 31));
        for (int64_t p = lo; p < hi; p++) out[p * W + (col >> 5)] &= msk;
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------
 * print_res (memo_query.py:65-71).
 *   conservation  print(*vec, sep='\n')  -> "v0\nv1\n...\n"; an empty vector
 *                 still prints the terminating "\n" (1-byte file)
 *   membership    np.savetxt(rec.astype('byte'), fmt='%i', delimiter=' ')
 *                 -> L lines of N "0"/"1" separated by one space; empty -> ""
 * Both return the number of bytes written into buf (cap must be large enough:
 * 6*L+1 resp. 2*N*L).
 * ---------------------------------------------------------------------- */
size_t oracle_emit_conservation(const uint16_t *vec, int64_t L, char *buf) {
    char *p = buf;
    if (L <= 0) { *p++ = '\n'; return 1; }
    for (int64_t i = 0; i < L; i++) p += sprintf(p, "%u\n", (unsigned)vec[i]);
    return (size_t)(p - buf);
}

size_t oracle_emit_membership(const uint32_t *bits, int64_t L, int64_t N, char *buf) {
    char *p = buf;
    int64_t W = (N + 31) / 32;
    for (int64_t i = 0; i < L; i++) {
        for (int64_t g = 0; g < N; g++) {
            *p++ = ((bits[i * W + (g >> 5)] >> (g & 31)) & 1u) ? '1' : '0';
            *p++ = (g + 1 < N) ? ' ' : '\n';
        }
        if (N == 0) *p++ = '\n';
    }
    return (size_t)(p - buf);
}

/* ------------------------------------------------------------------------
 * Synthetic pangenome index (DESIGN.md "Synthetic workload"; BASELINE.json
 * configs 2-5).  Index-addressable: row i depends only on (i, params), so the
 * CPU, one GPU and every shard of a multi-GPU run generate identical rows.
 *   start_i = 1 + floor(i * den / num)        num/den = rows per pivot position
 *   end_i   = start_i + (mix(seed, 2i)   mod 60)
 *   order_i = 1 +        (mix(seed, 2i+1) mod (N-1))
 * mix = splitmix64 finaliser of (seed + (x+1) * 0x9E3779B97F4A7C15).
 * ---------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t seed, uint64_t x) {
    uint64_t z = seed + (x + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_synth_rows(uint64_t row_begin, uint64_t count, uint64_t num, uint64_t den,
                       int64_t N, uint64_t seed, int64_t *s, int64_t *e, int64_t *o) {
    for (uint64_t j = 0; j < count; j++) {
        uint64_t i = row_begin + j;
        int64_t st = 1 + (int64_t)((i * den) / num);
        s[j] = st;
        e[j] = st + (int64_t)(mix64(seed, 2 * i) % 60);
        o[j] = 1 + (int64_t)(mix64(seed, 2 * i + 1) % (uint64_t)(N - 1));
    }
}

/* checksum used by full-size GPU tests: FNV-1a over the result bytes */
uint64_t oracle_fnv1a(const uint8_t *p, uint64_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
