"""Wavefront life times and per-phase cycles of genasm_lane_kernel from the stats line of `bench.py --stats`
(scrg_debug_stats, lanes_per_pair = 1).  usage: wave_life.py <stderr of bench.py --stats> <wavefronts of the launch>"""
import sys, ast
for l in open(sys.argv[1]):
    if l.startswith('stats'):
        d = ast.literal_eval(l.split(':',1)[1].strip())
        n_waves = int(sys.argv[2])
        first_start = (1<<62) - d['diag_fallbacks']; last_start = d['diag_rounds']; last_end = d['cycles_diag_dc']; first_end = (1<<62) - d['cycles_diag_tb']
        print("mean life %.3f ms; start spread %.3f ms; first end after first start %.3f ms; last end after first start %.3f ms" % (
            d['cycles_tb_loop']/n_waves/1e5, (last_start-first_start)/1e5, (first_end-first_start)/1e5, (last_end-first_start)/1e5))
        print({k: round(v) for k, v in d.items() if k.startswith('cyc_per_round')}, 'tb pass 1 per round', round(d['tb_macro_steps'] / max(1, d['rounds'])))
