import subprocess, sys, re
L='/opt/rocm/lib/llvm/bin/'
obj = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
subprocess.run([L+'llvm-objcopy','-O','binary','--only-section=.hip_fatbin',obj,'/tmp/kres.fat'],check=True)
out = subprocess.run([L+'clang-offload-bundler','--list','--type=o','--input=/tmp/kres.fat'],capture_output=True,text=True)
tgt=[l for l in out.stdout.split() if 'gfx950' in l][0]
subprocess.run([L+'clang-offload-bundler','--unbundle','--type=o','--input=/tmp/kres.fat','--targets='+tgt,'--output=/tmp/kres.co'],check=True)
md = subprocess.run([L+'llvm-readelf','--notes','/tmp/kres.co'],capture_output=True,text=True).stdout
blocks = md.split('  - .agpr_count')
rows=[]
for b in blocks[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    g = lambda k: re.search(r'\.'+k+r':\s+(\d+)', b)
    rows.append((name, g('vgpr_count').group(1), re.match(r':\s+(\d+)', b).group(1), g('sgpr_count').group(1), g('private_segment_fixed_size').group(1), g('group_segment_fixed_size').group(1)))
dem = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, d in zip(rows, dem):
    d = d.replace('void q3::','')
    d = re.sub(r'\(.*','',d)
    if pat and pat not in d: continue
    print(f"{d[:90]:90s} vgpr {r[1]:>4} agpr {r[2]:>4} sgpr {r[3]:>4} scratch {r[4]:>5} lds {r[5]:>6}")
