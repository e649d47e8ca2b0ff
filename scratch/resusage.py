import subprocess, re, sys
src = sys.argv[1] if len(sys.argv) > 1 else '/root/repo/onto_vae_amd/csrc/ontovae_kernels.hip'
out = subprocess.run(['/opt/rocm/bin/hipcc','-x','hip','--offload-arch=gfx950','-O3','-std=c++17','-c','-Rpass-analysis=kernel-resource-usage',src,'-o','/tmp/k.o'],capture_output=True,text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r'remark: ([A-Za-z ]+): (.+?) \[-Rpass', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        if cur: rows.append(cur)
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip().split('(')[0]}
    else:
        cur[k] = v
if cur: rows.append(cur)
seen = set()
for r in rows:
    if r['name'] in seen: continue
    seen.add(r['name'])
    print(f"{r['name'][:62]:62s} vgpr={r.get('VGPRs','?'):>4s} agpr={r.get('AGPRs','?'):>3s} sgpr={r.get('TotalSGPRs', r.get('SGPRs','?')):>4s} occ={r.get('Occupancy [waves/SIMD]','?'):>2s} lds={r.get('LDS Size [bytes/block]','?'):>6s} scratch={r.get('ScratchSize [bytes/lane]','?')}")
