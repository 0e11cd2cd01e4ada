"""Per-function resources of a gfx950 assembly listing (hipcc -save-temps): VGPRs, SGPRs, scratch bytes per
lane, code bytes, scratch stores / loads in the text -- kernels AND the out-of-line device functions they call.
    hipcc --offload-arch=gfx950 -O3 ... -save-temps=obj -c pw_kernels.hip -o /tmp/x.o
    python tests/tools/isa_functions.py /tmp/pw_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import subprocess
import sys


def main(path):
    txt = open(path).read()
    # a function: "<name>:" at column 0 ... ".Lfunc_endN:" followed by the "; Function info:" / kernel info comments
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z[\w$.]+):\s*(?:;.*)?$", txt, re.M)]
    rows = []
    for k, (pos, name) in enumerate(starts):
        end = starts[k + 1][0] if k + 1 < len(starts) else len(txt)
        body = txt[pos:end]
        g = lambda pat: (re.search(pat, body) or [None, "?"])[1]
        if "; NumVgprs" not in body:
            continue
        rows.append((name, g(r"; NumVgprs: (\d+)"), g(r"; NumSgprs: (\d+)"), g(r"; ScratchSize: (\d+)"),
                     g(r"; codeLenInByte = (\d+)"), len(re.findall(r"scratch_store", body)),
                     len(re.findall(r"scratch_load", body)), len(re.findall(r"v_writelane", body)),
                     len(re.findall(r"s_swappc|s_setpc", body))))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    print("%-90s %5s %5s %8s %8s %7s %7s %7s %6s" % ("function", "VGPR", "SGPR", "scratch", "code B", "sc.st", "sc.ld", "wlane", "calls"))
    for r, dn in zip(rows, names):
        dn = re.sub(r"\(anonymous namespace\)::", "", dn)
        dn = re.sub(r"^void ", "", dn)
        dn = re.sub(r"\(.*", "", dn)
        print("%-90s %5s %5s %8s %8s %7d %7d %7d %6d" % ((dn[:90],) + r[1:]))


if __name__ == "__main__":
    main(sys.argv[1])
