"""Registers / spills / LDS of the kernels in libgnnloop.so (from the code object's metadata notes; no GPU needed).
usage: python scripts/dev/kernel_regs.py [substring ...]"""
import os, re, subprocess, sys, tempfile
LLVM = '/opt/rocm/lib/llvm/bin'
so = os.environ.get('GNNKERAS_AMD_LIB', os.path.join(os.path.dirname(__file__), '..', '..', 'gnnkeras_amd', 'csrc', 'libgnnloop.so'))
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'k.co')
    subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', so, fat])
    subprocess.check_call([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'])
    notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
pats = sys.argv[1:]
for e in re.split(r'\n\s+- \.agpr_count:', notes)[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, e) or [None, '?'])[1]
    name = g('name')
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if pats and not any(p in dn for p in pats): continue
    print(f"{dn[:110]:110s} agpr {e.split()[0]:>3s} vgpr {g('vgpr_count'):>3s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}")
