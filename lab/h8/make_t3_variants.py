#!/usr/bin/env python3
"""Lab builds of conv_patch_t3_kernel with the stage's weight piece issued behind the first C2W_T3_PLACE pixel columns of its MFMA group
(each column = 4 MFMAs; 0 = right behind the barrier, the product schedule).  python lab/h8/make_t3_variants.py 0 2 4 6
STATE: a record; patches the product conv_patch3.hip by text (applies as long as the stage's `issue_ahead` / `mfmas` hook lines are unchanged); the generated
lab source is not tracked."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "climate2weather_amd/csrc/conv_patch3.hip")).read()
old = '''        } else {
            if (ahead) {
                issue_ahead();
                ahead = false;
            }
        }'''
assert src.count(old) == 1
src = src.replace(old, '''        } else {
            if (ahead && (NARROW || C2W_T3_PLACE == 0)) {
                issue_ahead();
                ahead = false;
            }
        }''')
old = '''        mfmas(IC3<KH>{}, [&](int n) {
        });'''
assert src.count(old) == 1
src = src.replace(old, '''        mfmas(IC3<KH>{}, [&](int n) {
            if (C2W_T3_PLACE > 0 && n == C2W_T3_PLACE && ahead) {
                __builtin_amdgcn_sched_barrier(0);
                issue_ahead();
                ahead = false;
                __builtin_amdgcn_sched_barrier(0);
            }
        });''')
lab = os.path.join(root, "lab/csrc/conv_patch3_place_lab.hip")
open(lab, "w").write("#ifndef C2W_T3_PLACE\n#define C2W_T3_PLACE 0\n#endif\n" + src)
pkg = os.path.join(root, "climate2weather_amd")
procs = []
for v in sys.argv[1:]:
    d = os.path.join(pkg, "build/alt/t3p_" + v)
    os.makedirs(d, exist_ok=True)
    procs.append((v, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(root, "include"),
                                       "-I" + os.path.join(pkg, "csrc"), "-DC2W_T3_PLACE=" + v, "-Rpass-analysis=kernel-resource-usage", "-c", lab, "-o", d + "/conv_patch3.o"],
                                      stderr=subprocess.PIPE, text=True)))
for v, pr in procs:
    err = pr.communicate()[1]
    if pr.returncode:
        print(err[-3000:]); sys.exit(1)
    lines = err.splitlines()
    for i, l in enumerate(lines):
        if "Function Name: " in l and "t3_kernel" in l and "ILi16Et" in l:
            print(v, l.split("t3_kernel")[1][:24], " ".join(x.split("remark: ")[-1].split(" [-R")[0].split(":0:")[-1].strip() for x in lines[i + 1:i + 9] if " VGPRs:" in x or "ScratchSize" in x))
    objs = [os.path.join(pkg, "build", f + ".o") for f in "conv_igemm conv_patch wgrad wgrad_patch pointwise attention attention_mfma sampler conv_center sources_digest".split()]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o",
                           os.path.join(pkg, "build/alt/libc2w_t3p_%s.so" % v), os.path.join(pkg, "build/alt/t3p_" + v, "conv_patch3.o")] + objs)
    print("built libc2w_t3p_%s.so" % v)
