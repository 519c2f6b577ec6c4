"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md §HBM prescribes) of
one bench.py command into a record of profiles/hbm_traffic.json: HBM bytes per launch of the dominant iteration kernel.

    python scripts/parse_pmc.py FETCH.csv WRITE.csv NODES ARCS STATE_DIM H1 OUT.json [CONST_BYTES_PER_NODE [KERNEL_SUBSTRING [TAKEN]]]

KERNEL_SUBSTRING: which kernel's rows to average (default k_state_fused; k_state_xwide for the 129 .. 256-wide kernel); TAKEN: a note
on when / on which commit the passes were taken (recorded next to the numbers).

CONST_BYTES_PER_NODE: what the kernel reads per node for the iteration-invariant part of the first layer: 4 H1 (the constant C,
default) or 128 (the XC variant: the node's 32 constant inputs).

gfx950 corrections (same guide): FETCH_SIZE tallies the 128-B requests of wide coalesced reads (16 B per lane) at 64 B, so
those bytes are doubled; WRITE_SIZE is exact for 16 B / lane streaming stores; other access widths are uncalibrated.  The
kernel reads its state rows 16 B per lane but its row pointers, source ids and constant term as dwords, and one counter
cannot tell them apart, so two bounds are recorded:
    upper = 2 x FETCH                       (every read tallied at half)
    lower = 2 x FETCH - dword_bytes         (the dword reads tallied in full; dword_bytes = their algorithmic byte count)
`hbm_bytes_per_launch` (what bench.py reports as roofline.traffic) is the upper bound.  Units of the counters are KiB.
Every record carries `library_source_hash` (gnnkeras_amd._native.source_hash(): the csrc/ sources + include/gnnloop.h the passes ran
on); bench.py prints a record's traffic only when the hash equals that of the sources it runs on."""
import csv, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnkeras_amd._native import source_hash
fetch_csv, write_csv = sys.argv[1], sys.argv[2]
n_nodes, n_arcs, d, h1 = int(float(sys.argv[3])), int(float(sys.argv[4])), int(sys.argv[5]), int(sys.argv[6])
out = sys.argv[7]


KSUB = sys.argv[9] if len(sys.argv) > 9 else 'k_state_fused'


def vals(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if KSUB in r['Kernel_Name'] and r['Counter_Name'] == counter]
    return [float(r['Counter_Value']) for r in rows], (rows[0]['Kernel_Name'] if rows else '')


f, kname = vals(fetch_csv, 'FETCH_SIZE')
w, _ = vals(write_csv, 'WRITE_SIZE')
kname = kname.replace('void gnn::', '').split('(')[0].replace(', ', ',')
raw_fetch = 1024.0 * statistics.mean(f)
write_b = 1024.0 * statistics.mean(w)
const_bytes = int(sys.argv[8]) if len(sys.argv) > 8 else 4 * h1
dword_bytes = n_arcs * 4 + n_nodes * (4 + const_bytes)         # source ids, row pointers, constant term / constant inputs
upper, lower = 2.0 * raw_fetch + write_b, 2.0 * raw_fetch - dword_bytes + write_b
algorithmic = n_arcs * (4 + 4 * d) + n_nodes * (4 + 8 * d + 4 * h1)
rec = {'kernel': kname, 'const_bytes_per_node': const_bytes, 'nodes': n_nodes, 'arcs': n_arcs, 'state_dim': d, 'launches': len(f),
       'FETCH_SIZE_KiB_raw_mean': statistics.mean(f), 'WRITE_SIZE_KiB_mean': statistics.mean(w),
       'write_bytes': write_b, 'hbm_bytes_per_launch': upper, 'bounds': [lower, upper],
       'algorithmic_bytes_per_launch': algorithmic, 'traffic_over_algorithmic': [lower / algorithmic, upper / algorithmic],
       'taken': sys.argv[10] if len(sys.argv) > 10 else None, 'library_source_hash': source_hash(),
       'note': 'upper: every read doubled (gfx950 tallies 128-B requests at 64 B for 16 B/lane reads); lower: the dword reads '
               '(row pointers, source ids, constant term) taken as tallied in full'}
data = {'records': []}
if os.path.exists(out):
    try:
        data = json.load(open(out))
        if 'records' not in data: data = {'records': []}
    except Exception:
        pass
data['records'] = [r for r in data['records'] if not (r['kernel'] == kname and r['nodes'] == n_nodes and r['arcs'] == n_arcs)] + [rec]
json.dump(data, open(out, 'w'), indent=1)
print(json.dumps(rec))
