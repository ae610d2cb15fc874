"""Build tools/_ab/dbgtree.so: the library with cycle stamps between the phases of k_tree_compose (block 0 of each
launch -> g_dbg, read back through gf_debug_read).  Development only; the source tree is left untouched."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gadfly_amd._lib as L
p = os.path.join(L.CSRC, 'gadfly_hip.hip')
orig = open(p).read()
s = orig
a = s.index('template <int NS>\n__global__ void __launch_bounds__(256) k_tree_compose(const TreeArgs A) {')
b = s.index('// The top of the down-sweep: the whole range starts from the zero state')
body = s[a:b]


def rep(x, y):
    global body
    assert body.count(x) == 1, (body.count(x), x)
    body = body.replace(x, y)


rep('    double *w = v1, *vv = v1 + 64, *g2v = v1 + 128, *tmpv = v1 + 192;\n',
    '    double *w = v1, *vv = v1 + 64, *g2v = v1 + 128, *tmpv = v1 + 192;\n    long long ts[16]; int nts = 0;\n'
    '#define MK() ts[nts++] = clock64()\n    MK();\n')
rep('    if (tid < 64) tmpv[tid] = SR.m[tid];\n    __syncthreads();\n',
    '    if (tid < 64) tmpv[tid] = SR.m[tid];\n    __syncthreads();\n    MK(); /*1 loads a*/\n')
rep('    cb_gauss_jordan<NS>(Au, v1, tid, n);', '    MK(); /*2 product a*/\n    cb_gauss_jordan<NS>(Au, v1, tid, n); MK(); /*3 GJ*/\n')
rep('    // c. AL = D Xbar1\n', '    MK(); /*4 reload G2, g2v*/\n    // c. AL = D Xbar1\n')
rep('    // d. M0 = Phi2 ;', '    MK(); /*5 product c*/\n    // d. M0 = Phi2 ;')
rep('    cb_load<NS>(M0, SR.Phi, tid);\n    __syncthreads();\n', '    cb_load<NS>(M0, SR.Phi, tid);\n    __syncthreads();\n    MK(); /*6 load Phi2*/\n')
rep('    if (in) {\n        double *Sr = SR.S;', '    MK(); /*7 two products d*/\n    if (in) {\n        double *Sr = SR.S;')
rep('    // e. AL = Phi1 ;', '    MK(); /*8 S rmw*/\n    // e. AL = Phi1 ;')
rep('    cb_load<NS>(Au, SL.Phi, tid, LA);\n    __syncthreads();\n', '    cb_load<NS>(Au, SL.Phi, tid, LA);\n    __syncthreads();\n    MK(); /*9 load Phi1*/\n')
rep('    // f. M1 <- G2 (D Phi1)', '    MK(); /*10 two products e + store Phi*/\n    // f. M1 <- G2 (D Phi1)')
rep('    if (in) {\n        double *Gr = SR.G;', '    MK(); /*11 two products f + m*/\n    if (in) {\n        double *Gr = SR.G;')
body = body.rstrip()
assert body.endswith('}')
body = (body[:-1] + '    MK(); /*12 G store*/\n    if (tid == 0 && blockIdx.x == 0) for (int q = 0; q < nts; ++q) '
        'g_dbg[q] = (double)(ts[q] - ts[0]);\n}\n\n')
body = '__device__ double g_dbg[16];\n' + body
s = s[:a] + body + s[b:]
s += '''
extern "C" __attribute__((visibility("default"))) int gf_debug_read(double *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(double) * n);
}
'''
try:
    open(p, 'w').write(s)
    r = subprocess.run(L.hipcc_command("tools/_ab/dbgtree.so"), capture_output=True, text=True)
    print(r.stderr[-1500:])
finally:
    open(p, 'w').write(orig)
