"""Build tools/_ab/dbgfw.so: k_factorw with cycle stamps inside the sweep waves' row loop (sums over rows, wave 0 of
workgroup 0 -> g_dbgw, read back through gf_debug_read_w).  Development only."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gadfly_amd._lib as L
p = os.path.join(L.CSRC, 'gadfly_hip.hip')
orig = open(p).read()
s = orig
def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a[:60])
    s = s.replace(a, b)
rep('''    do {
        const int cur = (int)(n & 1), nxt = cur ^ 1;
        const double2 *pw = (const double2 *)sh.w[cur] + g, *pu = (const double2 *)sh.u[b0] + g;''',
    '''    do {
        const long long c0_ = clock64();
        const int cur = (int)(n & 1), nxt = cur ^ 1;
        const double2 *pw = (const double2 *)sh.w[cur] + g, *pu = (const double2 *)sh.u[b0] + g;''')
rep('''        // the serial part of the row wins the issue arbitration against the other workgroup's sweep
        __builtin_amdgcn_s_setprio(GF_CHAIN_PRIO);''','''        const long long c1_ = clock64();
        __builtin_amdgcn_s_setprio(GF_CHAIN_PRIO);''')
rep('''        const double p2 = read_lane(tmp, 47);       // column FCOL of the last sweep wave: u~ . F~
        *reinterpret_cast<double2 *>(sh.p[nxt][wave]) = double2{p1, p2};    // (uniform, every lane: no branch)
        wg_lds_barrier();''','''        const double p2 = read_lane(tmp, 47);       // column FCOL of the last sweep wave: u~ . F~
        const long long c2_ = clock64();
        *reinterpret_cast<double2 *>(sh.p[nxt][wave]) = double2{p1, p2};    // (uniform, every lane: no branch)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const long long c3_ = clock64();
        wg_lds_barrier();
        const long long c4_ = clock64();''')
rep('''        stop = (n & (WIDE_FAIL_CHECK - 1)) == WIDE_FAIL_CHECK - 1 && fail;
        ++n;
    } while (n < rows && !(de >= 0.0) && !stop);''','''        stop = (n & (WIDE_FAIL_CHECK - 1)) == WIDE_FAIL_CHECK - 1 && fail;
        ++n;
        const long long c5_ = clock64();
        ts_[0] += c1_ - c0_; ts_[1] += c2_ - c1_; ts_[2] += c3_ - c2_; ts_[3] += c4_ - c3_; ts_[4] += c5_ - c4_; ts_[5] += 1;
    } while (n < rows && !(de >= 0.0) && !stop);''')
rep('''    int64_t n = 0;
    bool stop = false;
    while (n < rows && !stop) {
    if (de >= 0.0) {                        // workgroup-uniform reset row''','''    int64_t n = 0;
    bool stop = false;
    long long ts_[6] = {0, 0, 0, 0, 0, 0};
    long long trs_ = 0, nrs_ = 0;
    const long long t00_ = clock64();
    while (n < rows && !stop) {
    const long long r0_ = clock64();
    if (de >= 0.0) {                        // workgroup-uniform reset row
        nrs_ += 1;''')
rep('''        q0 = 0.0;
        q1 = 0.0;
    }
    do {
        const long long c0_ = clock64();''','''        q0 = 0.0;
        q1 = 0.0;
    }
    trs_ += clock64() - r0_;
    do {
        const long long c0_ = clock64();''')
rep('''    __builtin_amdgcn_s_setprio(0);
    if (fail) return;                               // (the generator wave records the row)
    const int fin = (int)(rows & 1);                // buffer written by the last row''','''    __builtin_amdgcn_s_setprio(0);
    if (blockIdx.x == 0 && wave == 0 && lane == 0) { for (int q = 0; q < 6; ++q) g_dbgw[q] = (double)ts_[q]; g_dbgw[6] = (double)(clock64() - t00_); g_dbgw[7] = (double)trs_; g_dbgw[5] = (double)ts_[5] + 1e-6 * (double)nrs_; }
    if (fail) return;                               // (the generator wave records the row)
    const int fin = (int)(rows & 1);                // buffer written by the last row''')
rep('''// rows between looks at the failure flag (a non-positive pivot is recorded without a branch; the rows''','''__device__ double g_dbgw[8];
// rows between looks at the failure flag (a non-positive pivot is recorded without a branch; the rows''')
s += '''
extern "C" __attribute__((visibility("default"))) int gf_debug_read_w(double *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbgw), sizeof(double) * n);
}
'''
try:
    open(p, 'w').write(s)
    r = subprocess.run(L.hipcc_command("tools/_ab/dbgfw.so"), capture_output=True, text=True)
    print(r.stderr[-1500:])
finally:
    open(p, 'w').write(orig)
