"""Build tools/_ab/dbgcorr.so: k_corr_small with cycle stamps between its phases (development only)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gadfly_amd._lib as L
p = os.path.join(L.CSRC, 'gadfly_dense.hip')
orig = open(p).read()
s = orig
marks = [
 '    // ---- e = Y - X m (X symmetric, full rows are there)\n',
 '    // ---- X = R R^T with diagonal pivoting, on registers',
 '    // ---- G -> A (whole rows), w1 = G e\n',
 '    // ---- M = I - R^T T (rank x rank; identity beyond), then M -> A\n',
 '    // ---- M eliminated without pivoting',
 '    // ---- the VALUES come from an elimination of A = I - X G',
 '    bool pivoted = lane >= n;',
 '    // ---- row `lane` solved variable myk: v(myk)',
]
for q, txt in enumerate(marks):
    assert txt in s, txt
    s = s.replace(txt, '    GF_MARK(%d);\n' % q + txt, 1)
a = s.index('k_corr_small(const int P')
end = s.index('__global__ void __launch_bounds__(256)\nk_corr_finish_small(')
body = s[a:end].rstrip()
assert body.endswith('}')
body = body[:-1] + '''    GF_MARK(8);
    if (tid == 0) { double *dbg = quad_out + (size_t)(gridDim.x / count) * P + 16 * blockIdx.x; for (int q = 0; q < 9; ++q) dbg[q] = (double)(gf_ts[q] - gf_t0); dbg[9] = rank; }
}

'''
s = s[:a] + body + s[end:]
s = s.replace('    const double nan = __longlong_as_double(0x7ff8000000000000LL);\n    // ---- X -> A, Y, m; R := 0',
              '    const double nan = __longlong_as_double(0x7ff8000000000000LL);\n    long long gf_ts[9]; const long long gf_t0 = clock64();\n#define GF_MARK(q) gf_ts[q] = clock64()\n    // ---- X -> A, Y, m; R := 0', 1)
# the early return of a failed check would skip the stamps: fine (not taken in the timing runs)
try:
    open(p, 'w').write(s)
    r = subprocess.run(L.hipcc_command("tools/_ab/dbgcorr.so"), capture_output=True, text=True)
    print(r.stderr[-1500:])
finally:
    open(p, 'w').write(orig)
