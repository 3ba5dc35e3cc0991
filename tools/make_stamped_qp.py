"""Diagnostic build of the QP kernel with s_memtime stamps at the /*@S:n*/ markers (never the product:
the stamped library is written to scratch/ and loaded only by scratch/run_dbg.py)."""
import os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "ihm2_amd", "csrc"), os.path.join(root, "scratch", "csrc_dbg")
shutil.rmtree(dst, ignore_errors=True); shutil.copytree(src, dst)
for f in os.listdir(dst):
    if f.endswith(".o"): os.remove(os.path.join(dst, f))
p = os.path.join(dst, "kernels_qp.hip"); s = open(p).read()
NS = 32
s = s.replace('#include "ihm2mpc_internal.h"', '#include "ihm2mpc_internal.h"\n#include <cstdio>', 1)
assert "    double car_L, car_W;\n" in s
s = s.replace("    double car_L, car_W;\n", "    double car_L, car_W;\n    long long *dbg;\n", 1)
s = s.replace("    // ---- LDS carve-up (doubles) ----", "    long long T[%d]; for (int q = 0; q < %d; q++) T[q] = 0; long long t_prev = __builtin_readcyclecounter(); int cur_sec = %d;\n#if QP_SET == 0 || defined(STAMP_SOFT)\n#define STAMP(i) do { long long t_now = __builtin_readcyclecounter(); T[cur_sec] += t_now - t_prev; t_prev = t_now; cur_sec = (i); } while (0)\n#else\n#define STAMP(i) do { (void)t_prev; (void)cur_sec; } while (0)   /* the soft / track-row set is built unstamped */\n#endif\n    // ---- LDS carve-up (doubles) ----" % (NS, NS, NS - 1))
s = re.sub(r"/\*@S:(\d+)\*/", lambda m: "STAMP(%s);" % m.group(1), s)
assert "        a.status[b] = st; a.qp_iter[b] = it;\n" in s
s = s.replace("        a.status[b] = st; a.qp_iter[b] = it;\n", "        a.status[b] = st; a.qp_iter[b] = it;\n        STAMP(%d);\n        if (b < 4) { for (int q = 0; q < %d; q++) a.dbg[b * %d + q] = T[q]; a.dbg[b * %d + %d] = it; }\n" % (NS - 2, NS, NS + 1, NS + 1, NS))
s = s.replace("    a.lin = h->lin;", "    static long long *dbg = nullptr; if (!dbg) (void)hipMalloc((void**)&dbg, 4 * %d * sizeof(long long)); a.dbg = dbg;\n    a.lin = h->lin;" % (NS + 1))
assert "#undef LAUNCH_QP\n    return 0;" in s
s = s.replace("#undef LAUNCH_QP\n    return 0;", "#undef LAUNCH_QP\n    { long long hb[4 * NSP]; (void)hipMemcpy(hb, a.dbg, sizeof hb, hipMemcpyDeviceToHost); static int cnt = 0; if (cnt++ % 10 == 5) for (int w = 0; w < 3; w++) { printf(\"[stamps b=%d it=%lld]\", w, hb[w * NSP + NSP - 1]); for (int q = 0; q < NSP - 1; q++) printf(\" %lld\", hb[w * NSP + q]); printf(\"\\n\"); } }\n    return 0;".replace("NSP", str(NS + 1)))
open(p, "w").write(s)
mk = os.path.join(dst, "Makefile"); m = open(mk).read().replace("OUT     = ../libihm2mpc.so", "OUT     = libihm2mpc_dbg.so")
if os.environ.get("STAMP_SOFT"): m = m.replace("-DQP_SET=1", "-DQP_SET=1 -DSTAMP_SOFT")      # stamps in the soft / track-row set as well
open(mk, "w").write(m)
subprocess.check_call(["make", "-C", dst, "-j4", "-s"])
print("built", os.path.join(dst, "libihm2mpc_dbg.so"))
