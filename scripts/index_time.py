"""Time the synthetic-genome + index build at a given size (Mbp) on this box: python scripts/index_time.py 3100"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bwa-mem-gpu_amd"))
import tools_py as tp
mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 3100
print("cpus", os.cpu_count(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "-",
      "affinity", len(os.sched_getaffinity(0)), flush=True)
print(subprocess.run("free -g | head -2; df -h /dev/shm | tail -1; nproc; lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket'", shell=True, capture_output=True, text=True).stdout, flush=True)
lens = tp.contig_lengths(mbp * 1000000)
d = "/dev/shm/bwahip_idx_time"; os.makedirs(d, exist_ok=True)
t0 = time.time(); g = tp.make_genome(38, lens, True); print(f"genome {time.time()-t0:.1f}s", flush=True)
t0 = time.time(); tp.write_fasta(f"{d}/g.fa", g, lens); print(f"fasta {time.time()-t0:.1f}s", flush=True)
del g
t0 = time.time(); subprocess.check_call([os.path.join(ROOT, "bwa-mem-gpu_amd", "tools", "mkindex"), f"{d}/g.fa", f"{d}/g"], env=dict(os.environ, MKINDEX_VERBOSE="1")); print(f"mkindex {time.time()-t0:.1f}s", flush=True)
print(subprocess.run(f"ls -la {d}; free -g | head -2", shell=True, capture_output=True, text=True).stdout)
import shutil; shutil.rmtree(d)
