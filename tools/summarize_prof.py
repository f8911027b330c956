"""Turn rocprofv3 result databases (gpurun_out/<dir>/*.db) into the small summaries kept under profiles/.
usage: python tools/summarize_prof.py <prof_dir> <tag>     (expects stats_results.db, fetch_results.db, write_results.db)"""
import json, os, sqlite3, sys

src, tag = sys.argv[1], sys.argv[2]
batch = sys.argv[3] if len(sys.argv) > 3 else '16'
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
os.makedirs(out, exist_ok=True)

def q(db, sql):
    return list(sqlite3.connect(os.path.join(src, db)).cursor().execute(sql))

rows = q('stats_results.db', "select name,total_calls,total_duration,average,percentage from top_kernels")
with open(os.path.join(out, '%s_kernel_stats.csv' % tag), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (batch %s x 512x512, 5-level); durations in us\n' % batch)
    f.write('name,calls,total_us,avg_us,percent\n')
    for r in rows:
        f.write('"%s",%d,%.3f,%.3f,%.3f\n' % r)

pmc = {}
for db, ctr in (('fetch_results.db', 'FETCH_SIZE'), ('write_results.db', 'WRITE_SIZE')):
    for name, n, avg in q(db, "select kernel_name, count(*), avg(value) from counters_collection where counter_name='%s' group by kernel_name" % ctr):
        pmc.setdefault(name, {})[ctr + '_KB_per_launch'] = avg
        pmc[name]['launches_' + ctr] = n
with open(os.path.join(out, '%s_pmc_hbm.csv' % tag), 'w') as f:
    f.write('# separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --steps 1 --warmup 1 --no-prof (batch %s x 512x512)\n' % batch)
    f.write('# raw counter averages per launch in KB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane)\n')
    f.write('# coalesced streaming read -> hbm_read_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is uncalibrated (taken as is).\n')
    f.write('kernel,launches,FETCH_SIZE_KB,WRITE_SIZE_KB,corrected_hbm_MB_per_launch\n')
    for name, d in sorted(pmc.items(), key=lambda kv: -(kv[1].get('FETCH_SIZE_KB_per_launch', 0) * kv[1].get('launches_FETCH_SIZE', 0))):
        fe, wr = d.get('FETCH_SIZE_KB_per_launch', 0.0), d.get('WRITE_SIZE_KB_per_launch', 0.0)
        f.write('"%s",%d,%.1f,%.1f,%.3f\n' % (name, d.get('launches_FETCH_SIZE', 0), fe, wr, (2 * fe + wr) * 1024 / 1e6))

# conv3x3 class: launch-weighted average over the generic template instantiations (<..., true> = conv1_1 inside the patch
# loader is bench.py's class conv12, not part of the class the roofline is quoted on)
tot_n = tot_f = tot_w = 0
for name, d in pmc.items():
    if 'conv3x3_mfma_kernel' in name and 'true>' not in name:
        n = d.get('launches_FETCH_SIZE', 0)
        tot_n += n
        tot_f += n * d.get('FETCH_SIZE_KB_per_launch', 0.0)
        tot_w += d.get('launches_WRITE_SIZE', 0) * d.get('WRITE_SIZE_KB_per_launch', 0.0)
summary = {'workload': 'batch %s x 512x512, 5-level' % batch, 'kernel_class': 'conv3x3_mfma_kernel', 'launches': tot_n,
           'fetch_size_KB_per_launch': tot_f / max(1, tot_n), 'write_size_KB_per_launch': tot_w / max(1, tot_n),
           'hbm_bytes_per_launch_corrected': (2 * tot_f + tot_w) * 1024 / max(1, tot_n),
           'correction': '2 x FETCH_SIZE (gfx950 wide-read undercount, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KB -> bytes'}
json.dump(summary, open(os.path.join(out, '%s_pmc_conv3x3.json' % tag), 'w'), indent=1)
print(json.dumps(summary))
