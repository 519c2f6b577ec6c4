"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md §HBM prescribes)
into profiles/hbm_traffic.json: HBM bytes per launch of the fused iteration kernel.
gfx950 corrections (same guide): FETCH_SIZE counts 128-B requests at 64 B for wide coalesced (16 B / lane) reads, so it
is doubled; WRITE_SIZE is exact for 16 B / lane streaming stores. Units are KiB."""
import csv, json, sys, statistics
fetch_csv, write_csv, n_nodes, n_arcs, out = sys.argv[1], sys.argv[2], int(float(sys.argv[3])), int(float(sys.argv[4])), sys.argv[5]
def vals(path, counter):
    return [float(r['Counter_Value']) for r in csv.DictReader(open(path))
            if 'k_state_fused' in r['Kernel_Name'] and r['Counter_Name'] == counter]
f, w = vals(fetch_csv, 'FETCH_SIZE'), vals(write_csv, 'WRITE_SIZE')
fetch_b = 2.0 * 1024.0 * statistics.mean(f)
write_b = 1024.0 * statistics.mean(w)
res = {'workload_nodes': n_nodes, 'workload_arcs': n_arcs, 'launches': len(f),
       'FETCH_SIZE_KiB_raw_mean': statistics.mean(f), 'WRITE_SIZE_KiB_mean': statistics.mean(w),
       'fetch_bytes_corrected_x2': fetch_b, 'write_bytes': write_b, 'hbm_bytes_per_launch': fetch_b + write_b,
       'note': 'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for 16 B/lane reads); '
               'dword index loads are a small uncalibrated share'}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
