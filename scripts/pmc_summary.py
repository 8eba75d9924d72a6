# gpurun_out/pmc_<COUNTER>/**/counter_collection.csv (one rocprofv3 --pmc pass per counter, scripts/gpu_pmc.sh) ->
# gpurun_out/<round>_pmc_traffic.json (copy it to profiles/): per kernel family of the train step the HBM bytes per launch
# (FETCH_SIZE is tallied at half for wide streaming reads on gfx950: doubled, MI355X_MICROARCH.md "HBM"; WRITE_SIZE as read;
# both in KiB), MFMA-busy cycles and GUI-active cycles, next to the algorithmic bytes of the metric-M shapes.
import csv, glob, json, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
FAMILIES = {'lstm_fwd': 'lstm_fwd_kernel', 'lstm_bwd': 'lstm_bwd_kernel', 'dec_persist_fwd': 'dec_persist_fwd_',
            'dec_persist_bwd': 'dec_persist_bwd_kernel', 'gemm_nt': 'gemm_nt_ring_kernel', 'gemm_tn_lstm': 'gemm_tn_ring_kernel', 'gemm_tn': 'gemm_tn_tr_kernel'}
B, T, H = 64, 800, 256
ALGO = {   # bytes per launch at the layer-1 shape (B=64, T=800, both directions)
    'lstm_fwd': B * T * 2 * (4 * H * 4 + 4 * H * 4 + H * 4 + H * 2),       # xproj read, gates written, c written, y (bf16) written
    'lstm_bwd': B * T * 2 * (4 * H * 4 + H * 4 + H * 4 + 4 * H * 2),       # gates, c, dy read; dz (bf16) written
}
per = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'):
    fs = glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % c, recursive=True)
    if not fs:
        print(c, 'no counter file')
        continue
    import os
    fs.sort(key=os.path.getmtime)          # several runs may have been merged into the directory: the newest one
    for r in csv.DictReader(open(fs[-1])):
        name = r['Kernel_Name']
        for fam, pat in FAMILIES.items():
            if pat in name:
                per.setdefault(fam, {}).setdefault(c, []).append(float(r['Counter_Value']))
out = {'source': 'rocprofv3 --pmc <counter> --kernel-trace, one pass per counter, of `bench.py --steps 1 --warmup 0 --no-graph` '
                 '(scripts/gpu_pmc.sh); per family: max over its dispatches (the longest shape), FETCH_SIZE doubled '
                 '(MI355X_MICROARCH.md HBM: gfx950 tallies wide streaming reads at half), sizes in KiB', 'kernels': {}}
for fam, d in per.items():
    k = {'dispatches': len(d.get('FETCH_SIZE', []))}
    f, w = max(d.get('FETCH_SIZE', [0])), max(d.get('WRITE_SIZE', [0]))
    k['fetch_size_kib'], k['write_size_kib'] = f, w
    k['traffic_bytes_per_launch'] = int(2 * f * 1024 + w * 1024)
    if fam in ALGO:
        k['algorithmic_bytes_per_launch'] = ALGO[fam]
        k['traffic_over_algorithmic'] = round(k['traffic_bytes_per_launch'] / ALGO[fam], 3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
        k['sq_valu_mfma_busy_cycles'] = max(d['SQ_VALU_MFMA_BUSY_CYCLES'])
    if 'GRBM_GUI_ACTIVE' in d:
        k['grbm_gui_active_sum_over_8_xcd'] = max(d['GRBM_GUI_ACTIVE'])
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and len(d['SQ_VALU_MFMA_BUSY_CYCLES']) == len(d['GRBM_GUI_ACTIVE']):
            # the SAME dispatch in both passes (the passes replay the same launches in the same order): the one with the most MFMA
            # cycles.  (max of each list on its own once paired the longest launch's MFMA cycles with the GUI-active count of a
            # launch that had waited 1.6 s behind the process's start-up: 0.05 % busy.)
            i = max(range(len(d['SQ_VALU_MFMA_BUSY_CYCLES'])), key=lambda j: d['SQ_VALU_MFMA_BUSY_CYCLES'][j])
            k['grbm_gui_active_sum_over_8_xcd'] = d['GRBM_GUI_ACTIVE'][i]
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
            k['mfma_busy_fraction_of_chip'] = round(k['sq_valu_mfma_busy_cycles'] / (k['grbm_gui_active_sum_over_8_xcd'] / 8 * 1024), 4)
    out['kernels'][fam] = k
    print(fam, k)
# the build these counters belong to: bench.py compares it with the build it times and reports `traffic_stale`
import sys as _sys
_sys.path.insert(0, '.')
import bench as _bench
out['csrc_digest'] = _bench.csrc_digest()
json.dump(out, open('gpurun_out/%s_pmc_traffic.json' % rnd, 'w'), indent=1)
