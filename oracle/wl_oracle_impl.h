/*
 * oracle/wl_oracle_impl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Type-generic body of the CPU oracle.  Included twice by wl_oracle.c with
 *   T   = float / double
 *   FN(name) = name##_f32 / name##_f64
 *
 * Every function is a literal restatement, in plain C, of one function of the
 * reference (JuliaDSP/Wavelets.jl v0.10.1).  The file:line each one follows is
 * cited above it (paths relative to /root/reference/src).  The loops keep the
 * reference's 1-based loop variables; arrays are accessed through the A1()
 * macro (1-based -> 0-based) so that index expressions can be compared with
 * the Julia text term by term.  Arithmetic is written exactly as the reference
 * writes it (a + b*c, never fused: the file is compiled with
 * -ffp-contract=off), so Float32 results are bit-comparable with Julia's.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * into this code.
 */

#define A1(a, i) ((a)[(i) - 1])

/* ------------------------------------------------------------------------ */
/* Transforms/transforms_filter.jl:362-369  @filtermainloop                  */
static inline void FN(filtermainloop)(T *si, int silen, const T *b, T val)
{
    for (int j = 2; j <= silen; ++j)
        A1(si, j - 1) = A1(si, j) + A1(b, j) * val;
    A1(si, silen) = A1(b, silen + 1) * val;
}
/* Transforms/transforms_filter.jl:370-377  @filtermainloopzero              */
static inline void FN(filtermainloopzero)(T *si, int silen)
{
    for (int j = 2; j <= silen; ++j)
        A1(si, j - 1) = A1(si, j);
    A1(si, silen) = (T)0.0;
}

/* Transforms/transforms_filter.jl:387-433  filtdown!                        */
static void FN(filtdown)(const T *f, int flen, T *si,
                         T *out, long iout, long nout,
                         const T *x, long ix, long shift, int ss)
{
    long nx = nout << 1;
    int silen = flen - 1;
    for (int j = 1; j <= silen; ++j) A1(si, j) = (T)0.0;
    long istart = flen + (ss ? 1 : 0);
    long dsshift = (flen % 2 + (ss ? 1 : 0)) % 2;

    wlo_range r1, rin, r2;
    wlo_splitdownrangeper(istart, ix, nx, shift, &r1, &rin, &r2);

    for (long i = r1.lo; i <= r1.hi; ++i) {
        T xatind = A1(x, wlo_mod(i - 1 + shift, nx) + ix);
        FN(filtermainloop)(si, silen, f, xatind);
    }
    long ixsh = -1 + shift + ix;
    for (long i = rin.lo; i <= rin.hi; ++i) {
        T xatind = A1(x, i + ixsh);
        if ((i + dsshift) % 2 == 0 && i >= istart)
            A1(out, ((i - istart) >> 1) + iout) = A1(si, 1) + A1(f, 1) * xatind;
        FN(filtermainloop)(si, silen, f, xatind);
    }
    for (long i = r2.lo; i <= r2.hi; ++i) {
        T xatind = A1(x, wlo_mod(i - 1 + shift, nx) + ix);
        if ((i + dsshift) % 2 == 0 && i >= istart)
            A1(out, ((i - istart) >> 1) + iout) = A1(si, 1) + A1(f, 1) * xatind;
        FN(filtermainloop)(si, silen, f, xatind);
    }
}

/* Transforms/transforms_filter.jl:467-541  filtup!                          */
static void FN(filtup)(int add2out, const T *f, int flen, T *si,
                       T *out, long iout, long nout,
                       const T *x, long ix, long shift, int ss)
{
    long nx = nout >> 1;
    int silen = flen - 1;
    for (int j = 1; j <= silen; ++j) A1(si, j) = (T)0.0;
    long istart = flen - wlo_rem(shift, 2);
    long dsshift = (ss ? 1 : 0) % 2;
    long shift_h = wlo_fld2(shift);               /* shift>>1 (arithmetic)   */

    wlo_range r1, rin, r2;
    wlo_splituprangeper(istart, ix, nx, nout, shift, &r1, &rin, &r2);

    T xatind = (T)0.0;
    for (long i = r1.lo; i <= r1.hi; ++i) {
        if ((i + dsshift) % 2 == 0) {
            FN(filtermainloopzero)(si, silen);
        } else {
            long xindex = wlo_mod(((i - 1) >> 1) + shift_h, nx) + ix;
            xatind = A1(x, xindex);
            FN(filtermainloop)(si, silen, f, xatind);
        }
    }
    long ixsh = shift_h + ix;
    for (long i = rin.lo; i <= rin.hi; ++i) {
        if ((i + dsshift) % 2 == 0) {
            xatind = (T)0.0;
        } else {
            long xindex = ((i - 1) >> 1) + ixsh;
            xatind = A1(x, xindex);
        }
        if (i >= istart) {
            if (add2out)
                A1(out, (i - istart) + iout) += A1(si, 1) + A1(f, 1) * xatind;
            else
                A1(out, (i - istart) + iout) = A1(si, 1) + A1(f, 1) * xatind;
        }
        if ((i + dsshift) % 2 == 0)
            FN(filtermainloopzero)(si, silen);
        else
            FN(filtermainloop)(si, silen, f, xatind);
    }
    for (long i = r2.lo; i <= r2.hi; ++i) {
        if ((i + dsshift) % 2 == 0) {
            xatind = (T)0.0;
        } else {
            long xindex = wlo_mod(((i - 1) >> 1) + shift_h, nx) + ix;
            xatind = A1(x, xindex);
        }
        if (i >= istart) {
            if (add2out)
                A1(out, (i - istart) + iout) += A1(si, 1) + A1(f, 1) * xatind;
            else
                A1(out, (i - istart) + iout) = A1(si, 1) + A1(f, 1) * xatind;
        }
        if ((i + dsshift) % 2 == 0)
            FN(filtermainloopzero)(si, silen);
        else
            FN(filtermainloop)(si, silen, f, xatind);
    }
}

/* WT/wt_main.jl:172-183 makereverseqmfpair + Util/util_main.jl:30 mirror.
 * The Float64 taps are converted to T first (copyto! into Vector{T}), the
 * sign pattern is applied in T.                                             */
static void FN(makereverseqmfpair)(const double *qmf, int flen, int fw,
                                   T *scfilter, T *dcfilter)
{
    T h[WLO_MAXF], mir[WLO_MAXF];
    for (int i = 0; i < flen; ++i) h[i] = (T)qmf[i];
    for (int i = 0; i < flen; ++i) mir[i] = (i % 2 == 0) ? h[i] : (T)(h[i] * (T)-1);
    if (fw) {
        for (int i = 0; i < flen; ++i) scfilter[i] = h[flen - 1 - i];
        for (int i = 0; i < flen; ++i) dcfilter[i] = mir[i];
    } else {
        for (int i = 0; i < flen; ++i) scfilter[i] = h[i];
        for (int i = 0; i < flen; ++i) dcfilter[i] = mir[flen - 1 - i];
    }
}

/* Transforms/transforms_filter.jl:63-83  unsafe_dwt1level! (filter)         */
static void FN(f_dwt1level)(T *y, const T *x, long n, int flen, int fw,
                            const T *dcfilter, const T *scfilter, T *si)
{
    long l = 1;
    if (fw) {
        FN(filtdown)(dcfilter, flen, si, y, wlo_detailindex(n, l, 1), wlo_detailn(n, l), x, 1, -flen + 1, 1);
        FN(filtdown)(scfilter, flen, si, y, 1, wlo_detailn(n, l), x, 1, 0, 0);
    } else {
        FN(filtup)(0, scfilter, flen, si, y, 1, wlo_detailn(n, l - 1), x, 1, -flen + 1, 0);
        FN(filtup)(1, dcfilter, flen, si, y, 1, wlo_detailn(n, l - 1), x, wlo_detailindex(n, l, 1), 0, 1);
    }
}

/* Transforms/transforms_filter.jl:13-62  _dwt! 1-D filter                   */
static int FN(f_dwt1d)(T *y, const T *x, long n, const double *qmf, int flen, int L, int fw)
{
    if (L < 0) return WLO_EINVAL_L;
    if (!wlo_sufficientpoweroftwo(n, L)) return WLO_EINVAL_SIZE;
    if (y == x) return WLO_EALIAS;
    if (L == 0) { memcpy(y, x, (size_t)n * sizeof(T)); return 0; }

    T scfilter[WLO_MAXF], dcfilter[WLO_MAXF], si[WLO_MAXF];
    FN(makereverseqmfpair)(qmf, flen, fw, scfilter, dcfilter);
    T *snew = (T *)malloc((size_t)(L > 1 ? (n >> 1) : 1) * sizeof(T));
    const T *s = x;

    for (int it = 0; it < L; ++it) {
        long l = fw ? (it + 1) : (L - it);
        int last = (it == L - 1);
        if (fw) {
            FN(filtdown)(dcfilter, flen, si, y, wlo_detailindex(n, l, 1), wlo_detailn(n, l), s, 1, -flen + 1, 1);
            FN(filtdown)(scfilter, flen, si, y, 1, wlo_detailn(n, l), s, 1, 0, 0);
        } else {
            FN(filtup)(0, scfilter, flen, si, y, 1, wlo_detailn(n, l - 1), s, 1, -flen + 1, 0);
            FN(filtup)(1, dcfilter, flen, si, y, 1, wlo_detailn(n, l - 1), x, wlo_detailindex(n, l, 1), 0, 1);
        }
        if (!last) memcpy(snew, y, (size_t)wlo_detailn(n, fw ? l : l - 1) * sizeof(T));
        if (L > 1) s = snew;
    }
    free(snew);
    return 0;
}

/* Util/util_main.jl:281-296 stridedcopy! (both directions)                  */
static void FN(stridedcopy_in)(T *b, const T *a, long ia, long inca, long n)
{
    for (long i = 1; i <= n; ++i) A1(b, i) = A1(a, ia + (i - 1) * inca);
}
static void FN(stridedcopy_out)(T *b, long ib, long incb, const T *a, long n)
{
    for (long i = 1; i <= n; ++i) A1(b, ib + (i - 1) * incb) = A1(a, i);
}

/* Transforms/transforms_filter.jl:85-96 dwt_transform_strided!
 * idx0 is the 1-based linear index of line 1; consecutive lines are
 * idx_step apart (row_idx / plane_idx of transforms_main.jl:219-224).       */
static void FN(f_transform_strided)(T *y, const T *x, long msub, long nsub, long stride,
                                    long idx0, long idx_step, T *tmpvec, T *tmpvec2,
                                    int flen, int fw, const T *dcf, const T *scf, T *si)
{
    for (long i = 1; i <= msub; ++i) {
        long xi = idx0 + (i - 1) * idx_step;
        FN(stridedcopy_in)(tmpvec, x, xi, stride, nsub);
        FN(f_dwt1level)(tmpvec2, tmpvec, nsub, flen, fw, dcf, scf, si);
        FN(stridedcopy_out)(y, xi, stride, tmpvec2, nsub);
    }
}
/* Transforms/transforms_filter.jl:98-109 dwt_transform_cols!                */
static void FN(f_transform_cols)(T *y, const T *x, long msub, long nsub,
                                 long idx0, long idx_step, T *tmpvec,
                                 int flen, int fw, const T *dcf, const T *scf, T *si)
{
    for (long i = 1; i <= nsub; ++i) {
        long xi = idx0 + (i - 1) * idx_step;
        memcpy(tmpvec, &A1(x, xi), (size_t)msub * sizeof(T));
        FN(f_dwt1level)(&A1(y, xi), tmpvec, msub, flen, fw, dcf, scf, si);
    }
}

/* Transforms/transforms_filter.jl:113-188  _dwt! 2-D filter                 */
static int FN(f_dwt2d)(T *y, const T *x, long m, long n, const double *qmf, int flen, int L, int fw)
{
    if (L < 0) return WLO_EINVAL_L;
    if (!wlo_sufficientpoweroftwo(m, L) || !wlo_sufficientpoweroftwo(n, L)) return WLO_EINVAL_SIZE;
    if (y == x) return WLO_EALIAS;
    if (L == 0) { memcpy(y, x, (size_t)(m * n) * sizeof(T)); return 0; }

    T scf[WLO_MAXF], dcf[WLO_MAXF], si[WLO_MAXF];
    FN(makereverseqmfpair)(qmf, flen, fw, scf, dcf);
    long tl = (n << 1) > m ? (n << 1) : m;
    T *tmpbuffer = (T *)malloc((size_t)tl * sizeof(T));
    long row_stride = m, nsub, msub;
    if (fw) { nsub = n; msub = m; }
    else { nsub = n / (1L << (L - 1)); msub = m / (1L << (L - 1)); memcpy(y, x, (size_t)(m * n) * sizeof(T)); }
    const T *inputArray = x;

    for (int it = 0; it < L; ++it) {
        T *tmpvec = tmpbuffer, *tmpvec2 = tmpbuffer, *tmpvec3 = tmpbuffer + nsub;
        if (fw) {
            /* rows: row_idx(i,m) = i, stride m */
            FN(f_transform_strided)(y, inputArray, msub, nsub, row_stride, 1, 1, tmpvec2, tmpvec3, flen, fw, dcf, scf, si);
            if (it == 0) inputArray = y;
            /* columns: col_idx(i,m) = 1 + (i-1)*m */
            FN(f_transform_cols)(y, y, msub, nsub, 1, m, tmpvec, flen, fw, dcf, scf, si);
        } else {
            FN(f_transform_cols)(y, inputArray, msub, nsub, 1, m, tmpvec, flen, fw, dcf, scf, si);
            if (it == 0) inputArray = y;
            FN(f_transform_strided)(y, y, msub, nsub, row_stride, 1, 1, tmpvec2, tmpvec3, flen, fw, dcf, scf, si);
        }
        msub = fw ? msub >> 1 : msub << 1;
        nsub = fw ? nsub >> 1 : nsub << 1;
    }
    free(tmpbuffer);
    return 0;
}

/* Same 2-D algorithm as f_dwt2d (forward and inverse), with the two per-level line loops
 * (dwt_transform_strided! over rows, dwt_transform_cols! over columns) spread
 * over OpenMP threads, each with its own scratch (si, tmpvec).  The reference
 * itself is single-threaded; this variant exists so that the GPU tests can check the
 * 8192 x 8192 configuration element by element in seconds.  Results are identical to
 * f_dwt2d (lines are independent; tests/test_oracle_properties.py compares the bits). */
static int FN(f_dwt2d_fw_mt)(T *y, const T *x, long m, long n, const double *qmf, int flen, int L, int fw)
{
    if (L < 0) return WLO_EINVAL_L;
    if (!wlo_sufficientpoweroftwo(m, L) || !wlo_sufficientpoweroftwo(n, L)) return WLO_EINVAL_SIZE;
    if (y == x) return WLO_EALIAS;
    if (L == 0) { memcpy(y, x, (size_t)(m * n) * sizeof(T)); return 0; }
    T scf[WLO_MAXF], dcf[WLO_MAXF];
    FN(makereverseqmfpair)(qmf, flen, fw, scf, dcf);
    long tl = (n << 1) > m ? (n << 1) : m;
    long nsub, msub;
    if (fw) { nsub = n; msub = m; }
    else { nsub = n / (1L << (L - 1)); msub = m / (1L << (L - 1)); memcpy(y, x, (size_t)(m * n) * sizeof(T)); }
    const T *inputArray = x;
    for (int it = 0; it < L; ++it) {
        const T *in = inputArray;
#pragma omp parallel
        {
            T si[WLO_MAXF];
            T *tmpbuffer = (T *)malloc((size_t)tl * sizeof(T));
            if (fw) {
#pragma omp for schedule(static)
                for (long i = 1; i <= msub; ++i) {          /* rows (transforms_filter.jl:161-168) */
                    FN(stridedcopy_in)(tmpbuffer, in, i, m, nsub);
                    FN(f_dwt1level)(tmpbuffer + nsub, tmpbuffer, nsub, flen, fw, dcf, scf, si);
                    FN(stridedcopy_out)(y, i, m, tmpbuffer + nsub, nsub);
                }
#pragma omp for schedule(static)
                for (long i = 1; i <= nsub; ++i) {          /* columns (:169-172) */
                    long xi = 1 + (i - 1) * m;
                    memcpy(tmpbuffer, &A1(y, xi), (size_t)msub * sizeof(T));
                    FN(f_dwt1level)(&A1(y, xi), tmpbuffer, msub, flen, fw, dcf, scf, si);
                }
            } else {
#pragma omp for schedule(static)
                for (long i = 1; i <= nsub; ++i) {          /* columns first (:174-178) */
                    long xi = 1 + (i - 1) * m;
                    memcpy(tmpbuffer, &A1(in, xi), (size_t)msub * sizeof(T));
                    FN(f_dwt1level)(&A1(y, xi), tmpbuffer, msub, flen, fw, dcf, scf, si);
                }
#pragma omp for schedule(static)
                for (long i = 1; i <= msub; ++i) {          /* then rows (:179-183) */
                    FN(stridedcopy_in)(tmpbuffer, y, i, m, nsub);
                    FN(f_dwt1level)(tmpbuffer + nsub, tmpbuffer, nsub, flen, fw, dcf, scf, si);
                    FN(stridedcopy_out)(y, i, m, tmpbuffer + nsub, nsub);
                }
            }
            free(tmpbuffer);
        }
        inputArray = y;
        msub = fw ? msub >> 1 : msub << 1;
        nsub = fw ? nsub >> 1 : nsub << 1;
    }
    return 0;
}

/* Transforms/transforms_filter.jl:192-294  _dwt! 3-D filter                 */
static int FN(f_dwt3d)(T *y, const T *x, long m, long n, long d, const double *qmf, int flen, int L, int fw)
{
    if (L < 0) return WLO_EINVAL_L;
    if (!wlo_sufficientpoweroftwo(m, L) || !wlo_sufficientpoweroftwo(n, L) || !wlo_sufficientpoweroftwo(d, L))
        return WLO_EINVAL_SIZE;
    if (y == x) return WLO_EALIAS;
    if (L == 0) { memcpy(y, x, (size_t)(m * n * d) * sizeof(T)); return 0; }

    T scf[WLO_MAXF], dcf[WLO_MAXF], si[WLO_MAXF];
    FN(makereverseqmfpair)(qmf, flen, fw, scf, dcf);
    long tl = m; if ((n << 1) > tl) tl = n << 1; if ((d << 1) > tl) tl = d << 1;
    T *tmpbuffer = (T *)malloc((size_t)tl * sizeof(T));
    long row_stride = m, plane_stride = m * n, msub, nsub, dsub;
    if (fw) { msub = m; nsub = n; dsub = d; }
    else {
        long q = 1L << (L - 1);
        msub = m / q; nsub = n / q; dsub = d / q;
        memcpy(y, x, (size_t)(m * n * d) * sizeof(T));
    }
    const T *inputArray = x;

    for (int it = 0; it < L; ++it) {
        T *tmpcol = tmpbuffer, *tmprow = tmpbuffer, *tmprow2 = tmpbuffer + nsub;
        T *tmphei = tmpbuffer, *tmphei2 = tmpbuffer + dsub;
        if (fw) {
            /* planes: plane_idx(i,j,m) = i + (j-1)*m */
            for (long j = 1; j <= nsub; ++j)
                FN(f_transform_strided)(y, inputArray, msub, dsub, plane_stride, 1 + (j - 1) * m, 1, tmphei, tmphei2, flen, fw, dcf, scf, si);
            if (it == 0) inputArray = y;
            /* rows: row_idx(i,j,m,n) = i + (j-1)*n*m */
            for (long j = 1; j <= dsub; ++j)
                FN(f_transform_strided)(y, y, msub, nsub, row_stride, 1 + (j - 1) * n * m, 1, tmprow, tmprow2, flen, fw, dcf, scf, si);
            /* columns: col_idx(i,j,m,n) = 1 + (i-1)*m + (j-1)*n*m */
            for (long j = 1; j <= dsub; ++j)
                FN(f_transform_cols)(y, y, msub, nsub, 1 + (j - 1) * n * m, m, tmpcol, flen, fw, dcf, scf, si);
        } else {
            for (long j = 1; j <= dsub; ++j)
                FN(f_transform_cols)(y, inputArray, msub, nsub, 1 + (j - 1) * n * m, m, tmpcol, flen, fw, dcf, scf, si);
            if (it == 0) inputArray = y;
            for (long j = 1; j <= dsub; ++j)
                FN(f_transform_strided)(y, y, msub, nsub, row_stride, 1 + (j - 1) * n * m, 1, tmprow, tmprow2, flen, fw, dcf, scf, si);
            for (long j = 1; j <= nsub; ++j)
                FN(f_transform_strided)(y, y, msub, dsub, plane_stride, 1 + (j - 1) * m, 1, tmphei, tmphei2, flen, fw, dcf, scf, si);
        }
        msub = fw ? msub >> 1 : msub << 1;
        nsub = fw ? nsub >> 1 : nsub << 1;
        dsub = fw ? dsub >> 1 : dsub << 1;
    }
    free(tmpbuffer);
    return 0;
}

/* Transforms/transforms_filter.jl:301-359  _wpt! 1-D filter                 */
static int FN(f_wpt1d)(T *y, const T *x, long n, const double *qmf, int flen,
                       const unsigned char *tree, long ntree, int fw)
{
    if (y == x) return WLO_EALIAS;
    if (!wlo_isvalidtree(n, tree, ntree)) return WLO_EINVAL_TREE;
    if (!A1(tree, 1)) { memcpy(y, x, (size_t)n * sizeof(T)); return 0; }

    T scf[WLO_MAXF], dcf[WLO_MAXF], si[WLO_MAXF];
    FN(makereverseqmfpair)(qmf, flen, fw, scf, dcf);
    long ns = fw ? (n >> 1) : n;
    T *snew = (T *)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(T));

    int first = 1;
    int Lmax = wlo_maxtransformlevels(n);
    int L = Lmax;
    while (L > 0) {
        long ix = 1, k = 1;
        int Lfw = fw ? Lmax - L : L - 1;
        long nj = wlo_detailn(n, Lfw);
        long treeind = (1L << Lfw) - 1;
        while (ix <= n) {
            if (A1(tree, treeind + k)) {
                T *dy = &A1(y, ix);
                const T *dx;
                if (first) {
                    dx = &A1(x, ix);
                } else {
                    memcpy(snew, dy, (size_t)nj * sizeof(T));
                    dx = snew;
                }
                FN(f_dwt1level)(dy, dx, nj, flen, fw, dcf, scf, si);
            } else if (first) {
                memcpy(&A1(y, ix), &A1(x, ix), (size_t)nj * sizeof(T));
            }
            ix += nj;
            k += 1;
        }
        L -= 1;
        first = 0;
    }
    free(snew);
    return 0;
}

/* ======================================================================== */
/*                               LIFTING                                    */
/* ======================================================================== */

typedef struct {
    int is_update;                 /* steptype: 0 = Predict, 1 = Update      */
    int nc;
    int shift;
    T coef[WLO_MAXC];
} FN(lsstep);

/* Transforms/transforms_lifting.jl:13-25 makescheme                         */
static void FN(makescheme)(const wlo_scheme *sc, int fw, FN(lsstep) *stepseq, T *norm1, T *norm2)
{
    int n = sc->nsteps;
    const double *cf = sc->coefs;
    int off[WLO_MAXSTEPS];
    int o = 0;
    for (int i = 0; i < n; ++i) { off[i] = o; o += sc->ncoef[i]; }
    for (int i = 1; i <= n; ++i) {
        int j = fw ? i : n + 1 - i;
        FN(lsstep) *st = &stepseq[i - 1];
        st->is_update = sc->is_update[j - 1];
        st->nc = sc->ncoef[j - 1];
        st->shift = sc->shift[j - 1];
        for (int k = 0; k < st->nc; ++k)
            st->coef[k] = (T)(cf[off[j - 1] + k] * (fw ? -1.0 : 1.0));
    }
    *norm1 = (T)(fw ? sc->norm1 : 1.0 / sc->norm1);
    *norm2 = (T)(fw ? sc->norm2 : 1.0 / sc->norm2);
}

/* Util/util_main.jl:142-160 split! (in place, range 1:n, tmp)               */
static void FN(split_ip)(T *a, long n, T *tmp)
{
    if (n == 2) return;
    long nt = (n >> 2) + ((n >> 1) % 2);
    for (long i = 1; i <= nt; ++i) A1(tmp, i) = A1(a, i << 1);
    for (long i = 1; i <= (n >> 1); ++i) A1(a, i) = A1(a, ((i - 1) << 1) + 1);
    for (long i = 0; i <= nt - 1; ++i) A1(a, n - i) = A1(a, n - (i << 1));
    memcpy(&A1(a, (n >> 1) + 1), tmp, (size_t)nt * sizeof(T));
}
/* Util/util_main.jl:183-204 split! (out of place, strided source)           */
static void FN(split_oop)(T *b, const T *a, long ia, long inca, long n)
{
    if (n == 2) { A1(b, 1) = A1(a, ia); A1(b, 2) = A1(a, ia + inca); return; }
    long h = n >> 1, inca2 = inca << 1;
    for (long i = 1; i <= h; ++i) A1(b, i) = A1(a, ia + (i - 1) * inca2);
    long iainca = ia + inca, hp1 = h + 1;
    for (long i = h + 1; i <= n; ++i) A1(b, i) = A1(a, iainca + (i - hp1) * inca2);
}
/* Util/util_main.jl:216-234 merge! (in place)                               */
static void FN(merge_ip)(T *a, long n, T *tmp)
{
    if (n == 2) return;
    long nt = (n >> 2) + ((n >> 1) % 2);
    memcpy(tmp, &A1(a, (n >> 1) + 1), (size_t)nt * sizeof(T));
    for (long i = nt - 1; i >= 0; --i) A1(a, n - 2 * i) = A1(a, n - i);
    for (long i = n >> 1; i >= 1; --i) A1(a, ((i - 1) << 1) + 1) = A1(a, i);
    for (long i = nt; i >= 1; --i) A1(a, i << 1) = A1(tmp, i);
}
/* Util/util_main.jl:257-278 merge! (out of place, strided destination)      */
static void FN(merge_oop)(T *b, long ib, long incb, const T *a, long n)
{
    if (n == 2) { A1(b, ib) = A1(a, 1); A1(b, ib + incb) = A1(a, 2); return; }
    long h = n >> 1, incb2 = incb << 1;
    for (long i = 1; i <= h; ++i) A1(b, ib + (i - 1) * incb2) = A1(a, i);
    long ibincb = ib + incb, hp1 = h + 1;
    for (long i = h + 1; i <= n; ++i) A1(b, ibincb + (i - hp1) * incb2) = A1(a, i);
}

/* Transforms/transforms_lifting.jl:323-359 normalize! variants              */
static void FN(normalize_ip)(T *x, long half, long ns, T n1, T n2)
{
    for (long i = 1; i <= half; ++i) A1(x, i) *= n1;
    for (long i = half + 1; i <= ns; ++i) A1(x, i) *= n2;
}
static void FN(normalize_to_strided)(T *y, long iy, long incy, const T *x, long half, long ns, T n1, T n2)
{
    for (long i = 1; i <= half; ++i) A1(y, iy + (i - 1) * incy) = n1 * A1(x, i);
    for (long i = half + 1; i <= ns; ++i) A1(y, iy + (i - 1) * incy) = n2 * A1(x, i);
}
static void FN(normalize_from_strided)(T *y, const T *x, long ix, long incx, long half, long ns, T n1, T n2)
{
    for (long i = 1; i <= half; ++i) A1(y, i) = n1 * A1(x, ix + (i - 1) * incx);
    for (long i = half + 1; i <= ns; ++i) A1(y, i) = n2 * A1(x, ix + (i - 1) * incx);
}

/* Transforms/transforms_lifting.jl:437-451 lift_perboundary!                */
static void FN(lift_perboundary)(T *x, long half, const T *c, int nc, wlo_range ir, long rhsis, int is_update)
{
    for (long i = ir.lo; i <= ir.hi; ++i)
        for (long k = 1; k <= nc; ++k) {
            long idx = is_update ? wlo_mod1(i + k - 1 + rhsis, half)
                                 : wlo_mod1(i + k - 1 + rhsis - half, half) + half;
            A1(x, i) += A1(c, k) * A1(x, idx);
        }
}
/* Transforms/transforms_lifting.jl:455-483 lift_inbounds!
 * (the nc>3 branch of the reference indexes c[0] -- a latent bug never
 * reached by the shipped schemes; the oracle refuses nc>3 at the entry)     */
static void FN(lift_inbounds)(T *x, const T *c, int nc, wlo_range ir, long rhsis)
{
    if (nc == 1) {
        T c1 = A1(c, 1);
        for (long i = ir.lo; i <= ir.hi; ++i) A1(x, i) += c1 * A1(x, i + rhsis);
    } else if (nc == 2) {
        T c1 = A1(c, 1), c2 = A1(c, 2);
        long rhsisp1 = rhsis + 1;
        for (long i = ir.lo; i <= ir.hi; ++i)
            A1(x, i) += c1 * A1(x, i + rhsis) + c2 * A1(x, i + rhsisp1);
    } else if (nc == 3) {
        T c1 = A1(c, 1), c2 = A1(c, 2), c3 = A1(c, 3);
        long rhsisp1 = rhsis + 1, rhsisp2 = rhsis + 2;
        for (long i = ir.lo; i <= ir.hi; ++i)
            A1(x, i) += c1 * A1(x, i + rhsis) + c2 * A1(x, i + rhsisp1) + c3 * A1(x, i + rhsisp2);
    }
}
/* Transforms/transforms_lifting.jl:366-381 lift!                            */
static void FN(lift)(T *x, long half, const FN(lsstep) *st)
{
    wlo_range lhsr, irange, rhsr; long rhsis;
    wlo_getliftranges(half, st->nc, st->shift, st->is_update, &lhsr, &irange, &rhsr, &rhsis);
    FN(lift_perboundary)(x, half, st->coef, st->nc, lhsr, rhsis, st->is_update);
    FN(lift_inbounds)(x, st->coef, st->nc, irange, rhsis);
    FN(lift_perboundary)(x, half, st->coef, st->nc, rhsr, rhsis, st->is_update);
}

/* Transforms/transforms_lifting.jl:82-122 unsafe_dwt1level! (lifting)       */
static void FN(l_dwt1level)(T *y, long iy, long incy, int oopc, T *oopv, long ns_oopv,
                            int fw, const FN(lsstep) *stepseq, int nsteps, T norm1, T norm2, T *tmp)
{
    if (!oopc) oopv = y;               /* caller passes y already offset when !oopc */
    long ns = ns_oopv, half = ns >> 1;
    if (fw) {
        if (oopc) FN(split_oop)(oopv, y, iy, incy, ns); else FN(split_ip)(oopv, ns, tmp);
        for (int s = 0; s < nsteps; ++s) FN(lift)(oopv, half, &stepseq[s]);
        if (oopc) FN(normalize_to_strided)(y, iy, incy, oopv, half, ns, norm1, norm2);
        else FN(normalize_ip)(oopv, half, ns, norm1, norm2);
    } else {
        if (oopc) FN(normalize_from_strided)(oopv, y, iy, incy, half, ns, norm1, norm2);
        else FN(normalize_ip)(oopv, half, ns, norm1, norm2);
        for (int s = 0; s < nsteps; ++s) FN(lift)(oopv, half, &stepseq[s]);
        if (oopc) FN(merge_oop)(y, iy, incy, oopv, ns); else FN(merge_ip)(oopv, ns, tmp);
    }
}

/* Transforms/transforms_lifting.jl:30-76 _dwt! 1-D lifting (in place)       */
static int FN(l_dwt1d)(T *y, long n, const wlo_scheme *sc, int L, int fw)
{
    if (L < 0) return WLO_EINVAL_L;
    if (!wlo_sufficientpoweroftwo(n, L)) return WLO_EINVAL_SIZE;
    if (!wlo_scheme_ok(sc)) return WLO_EINVAL_SCHEME;
    if (L == 0) return 0;
    long ns = fw ? n : wlo_detailn(n, L - 1);
    long half = ns >> 1;
    FN(lsstep) stepseq[WLO_MAXSTEPS]; T norm1, norm2;
    FN(makescheme)(sc, fw, stepseq, &norm1, &norm2);
    T *tmp = (T *)malloc((size_t)((n >> 2) + 2) * sizeof(T));
    for (int it = 0; it < L; ++it) {
        if (fw) {
            FN(split_ip)(y, ns, tmp);
            for (int s = 0; s < sc->nsteps; ++s) FN(lift)(y, half, &stepseq[s]);
            FN(normalize_ip)(y, half, ns, norm1, norm2);
            ns >>= 1; half >>= 1;
        } else {
            FN(normalize_ip)(y, half, ns, norm1, norm2);
            for (int s = 0; s < sc->nsteps; ++s) FN(lift)(y, half, &stepseq[s]);
            FN(merge_ip)(y, ns, tmp);
            ns <<= 1; half <<= 1;
        }
    }
    free(tmp);
    return 0;
}

/* Transforms/transforms_lifting.jl:128-194 _dwt! 2-D lifting (square only)  */
static int FN(l_dwt2d)(T *y, long m, long n2, const wlo_scheme *sc, int L, int fw)
{
    long n = m;
    if (m != n2) return WLO_EINVAL_CUBE;
    if (L < 0) return WLO_EINVAL_L;
    if (!wlo_sufficientpoweroftwo(n, L)) return WLO_EINVAL_SIZE;
    if (!wlo_scheme_ok(sc)) return WLO_EINVAL_SCHEME;
    if (L == 0) return 0;
    long row_stride = n;
    long nsub = fw ? n : n / (1L << (L - 1));
    FN(lsstep) stepseq[WLO_MAXSTEPS]; T norm1, norm2;
    FN(makescheme)(sc, fw, stepseq, &norm1, &norm2);
    T *tmp = (T *)malloc((size_t)((n >> 2) + 2) * sizeof(T));
    T *tmpvec = (T *)malloc((size_t)n * sizeof(T));
    for (int it = 0; it < L; ++it) {
        if (fw) {
            for (long i = 1; i <= nsub; ++i)           /* rows: row_idx(i,n) = i */
                FN(l_dwt1level)(y, i, row_stride, 1, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
            for (long i = 1; i <= nsub; ++i)           /* columns: col_idx(i,n)  */
                FN(l_dwt1level)(&A1(y, 1 + (i - 1) * n), 1, 1, 0, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
        } else {
            for (long i = 1; i <= nsub; ++i)
                FN(l_dwt1level)(&A1(y, 1 + (i - 1) * n), 1, 1, 0, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
            for (long i = 1; i <= nsub; ++i)
                FN(l_dwt1level)(y, i, row_stride, 1, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
        }
        nsub = fw ? nsub >> 1 : nsub << 1;
    }
    free(tmp); free(tmpvec);
    return 0;
}

/* Transforms/transforms_lifting.jl:200-278 _dwt! 3-D lifting (cube only)    */
static int FN(l_dwt3d)(T *y, long m, long n2, long n3, const wlo_scheme *sc, int L, int fw)
{
    long n = m;
    if (m != n2 || m != n3) return WLO_EINVAL_CUBE;
    if (L < 0) return WLO_EINVAL_L;
    if (!wlo_sufficientpoweroftwo(n, L)) return WLO_EINVAL_SIZE;
    if (!wlo_scheme_ok(sc)) return WLO_EINVAL_SCHEME;
    if (L == 0) return 0;
    long row_stride = n, plane_stride = n * n;
    long nsub = fw ? n : n / (1L << (L - 1));
    FN(lsstep) stepseq[WLO_MAXSTEPS]; T norm1, norm2;
    FN(makescheme)(sc, fw, stepseq, &norm1, &norm2);
    T *tmp = (T *)malloc((size_t)((n >> 2) + 2) * sizeof(T));
    T *tmpvec = (T *)malloc((size_t)n * sizeof(T));
    for (int it = 0; it < L; ++it) {
        if (fw) {
            for (long i = 1; i <= nsub; ++i) for (long j = 1; j <= nsub; ++j)   /* planes */
                FN(l_dwt1level)(y, i + (j - 1) * n, plane_stride, 1, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
            for (long i = 1; i <= nsub; ++i) for (long j = 1; j <= nsub; ++j)   /* rows   */
                FN(l_dwt1level)(y, i + (j - 1) * n * n, row_stride, 1, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
            for (long i = 1; i <= nsub; ++i) for (long j = 1; j <= nsub; ++j)   /* cols   */
                FN(l_dwt1level)(&A1(y, 1 + (i - 1) * n + (j - 1) * n * n), 1, 1, 0, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
        } else {
            for (long i = 1; i <= nsub; ++i) for (long j = 1; j <= nsub; ++j)
                FN(l_dwt1level)(&A1(y, 1 + (i - 1) * n + (j - 1) * n * n), 1, 1, 0, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
            for (long i = 1; i <= nsub; ++i) for (long j = 1; j <= nsub; ++j)
                FN(l_dwt1level)(y, i + (j - 1) * n * n, row_stride, 1, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
            for (long i = 1; i <= nsub; ++i) for (long j = 1; j <= nsub; ++j)
                FN(l_dwt1level)(y, i + (j - 1) * n, plane_stride, 1, tmpvec, nsub, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
        }
        nsub = fw ? nsub >> 1 : nsub << 1;
    }
    free(tmp); free(tmpvec);
    return 0;
}

/* Transforms/transforms_lifting.jl:283-319 _wpt! 1-D lifting (in place)     */
static int FN(l_wpt1d)(T *y, long n, const wlo_scheme *sc, const unsigned char *tree, long ntree, int fw)
{
    if (!wlo_isvalidtree(n, tree, ntree)) return WLO_EINVAL_TREE;
    if (!wlo_scheme_ok(sc)) return WLO_EINVAL_SCHEME;
    if (!A1(tree, 1)) return 0;
    FN(lsstep) stepseq[WLO_MAXSTEPS]; T norm1, norm2;
    FN(makescheme)(sc, fw, stepseq, &norm1, &norm2);
    T *tmp = (T *)malloc((size_t)((n >> 2) + 2) * sizeof(T));
    int Lmax = wlo_maxtransformlevels(n);
    int L = Lmax;
    while (L > 0) {
        long ix = 1, k = 1;
        int Lfw = fw ? Lmax - L : L - 1;
        long nj = wlo_detailn(n, Lfw);
        long treeind = (1L << Lfw) - 1;
        while (ix <= n) {
            if (A1(tree, treeind + k))
                FN(l_dwt1level)(&A1(y, ix), 1, 1, 0, tmp, nj, fw, stepseq, sc->nsteps, norm1, norm2, tmp);
            ix += nj;
            k += 1;
        }
        L -= 1;
    }
    free(tmp);
    return 0;
}

#undef A1
