"""Dev aid: compile dpig_conv.hip with -save-temps and print a compact view of a kernel's hot loop."""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "disentangled-person-image-generation_amd", "csrc")
os.makedirs("/tmp/asm", exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(root, "include"), "-I", csrc,
                       "-save-temps", "-c", os.path.join(csrc, "dpig_conv.hip"), "-o", "/tmp/asm/conv.o"], cwd="/tmp/asm",
                      stderr=subprocess.DEVNULL)
s = open('/tmp/asm/dpig_conv-hip-amdgcn-amd-amdhsa-gfx950.s').read()
name = sys.argv[1] if len(sys.argv) > 1 else '_ZN4dpig18gather_gemm_kernelILb0ELb1EEEvNS_8GGParamsE'
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
lines = s[i:j].split('\n')
# innermost loop = region between the label that is target of the last backward branch containing mfma
labels = {l.split(':')[0]: k for k, l in enumerate(lines) if re.match(r'\.LBB\d+_\d+:', l)}
best = None
for k, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < k and any('mfma' in x for x in lines[labels[t]:k]):
            if best is None or (k - labels[t]) < (best[1] - best[0]): best = (labels[t], k)
lo, hi = best
ops = [l.split()[0] for l in lines[lo:hi + 1] if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
print(len(ops), collections.Counter(ops).most_common(14))
seq = []
for o in ops:
    k = 'M' if 'mfma' in o else ('GL' if o.startswith(('global_load', 'buffer_load')) else ('DR' if o.startswith('ds_read') else ('DW' if o.startswith('ds_write') else ('W' if o == 's_waitcnt' else ('BAR' if 'barrier' in o else ('BR' if 'branch' in o else 'x'))))))
    if seq and seq[-1][0] == k: seq[-1][1] += 1
    else: seq.append([k, 1])
print(' '.join('%s%d' % (k, n) for k, n in seq))
