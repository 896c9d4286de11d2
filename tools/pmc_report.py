import csv, collections, glob, os, sys
for name in sys.argv[1:]:
    f = max(glob.glob(f"gpurun_out/pmc_{name}/*/*counter_collection.csv"), key=os.path.getmtime)
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if any(x in r['Kernel_Name'] for x in ('cost_edges', 'heuristic', 'pose_sweep', 'cover_sweep', 'k_sweeps', 'solve_edges', 'plan_skips', 'approach_events', 'cover_finish', 'deferred_list', 'near_events')):
            d[r['Kernel_Name'][:22]][r['Counter_Name']] = d[r['Kernel_Name'][:22]].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for k, v in d.items():
        w = v['SQ_WAVES']
        print(f"{name:12s} {k:24s} waves {w:8.0f} VALU/wave {v['SQ_INSTS_VALU']/w:8.0f} SALU/wave {v['SQ_INSTS_SALU']/w:7.0f} wave_cyc {4*v['SQ_WAVE_CYCLES']/w:9.0f} active_valu {4*v['SQ_ACTIVE_INST_VALU']/w:8.0f}")
