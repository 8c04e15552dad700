"""Wavefront life times and per-phase cycles of genasm_lane_kernel from the stats line of `bench.py --stats`
(Aligner.debug_stats_lane).  usage: wave_life.py <stderr of bench.py --stats> <wavefronts of the launch>"""
import sys, ast
for l in open(sys.argv[1]):
    if l.startswith('stats'):
        d = ast.literal_eval(l.split(':', 1)[1].strip())
        n_waves = int(sys.argv[2])
        t0 = d['first_start']
        print("mean life %.3f ms; start spread %.3f ms; first end after first start %.3f ms; last end after first start %.3f ms" % (
            d['life_ticks_sum'] / n_waves / 1e5, (d['last_start'] - t0) / 1e5, (d['first_end'] - t0) / 1e5, (d['last_end'] - t0) / 1e5))
        print({k: round(v) for k, v in d.items() if k.startswith('cyc_per_round')}, 'rounds', d['rounds'], 'short-text rounds', d['short_text_rounds'])
