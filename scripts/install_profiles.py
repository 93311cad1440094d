"""Copies what scripts/collect_profiles.sh left under gpurun_out/prof/<MODEL>/ into profiles/<round>_<MODEL>_*: the kernel
statistics cut down to the engine's own kernels (torch's fill / copy kernels of the corpus generator dropped), the bench line
printed under the profiler and the counter summary.  usage: install_profiles.py <round tag, e.g. r02> <MODEL> [<MODEL> ...]"""
import csv
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINE_KERNELS = ("sgd_", "sampler", "score_", "screen", "correct_clicked", "build_queries", "item_consts", "query_items", "project_rows",
                  "part_", "mark_hot", "apply_global", "fold_", "sample_count", "topk_", "gather_rows", "shard_")


def main():
    tag, models = sys.argv[1], sys.argv[2:]
    for model in models:
        src = os.path.join(ROOT, "gpurun_out", "prof", model)
        rows = list(csv.reader(open(os.path.join(src, "stats", "s_kernel_stats.csv"))))
        keep = [r for r in rows[1:] if any(k in r[0] for k in ENGINE_KERNELS)]
        with open(os.path.join(ROOT, "profiles", f"{tag}_{model}_kernel_stats.csv"), "w") as f:
            f.write(",".join('"%s"' % c for c in rows[0]) + "\n")
            for r in keep:
                f.write('"%s",' % r[0].replace('"', '""') + ",".join(r[1:]) + "\n")
        shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(ROOT, "profiles", f"{tag}_{model}_bench_under_rocprof.json"))
        shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(ROOT, "profiles", f"{tag}_{model}_pmc_summary.json"))
        print(model, len(keep), "kernels")
        for r in keep[:6]:
            print("   %-60s calls %5s  avg %.3f ms" % (r[0][:60], r[1], float(r[3]) / 1e6))


if __name__ == "__main__":
    main()
