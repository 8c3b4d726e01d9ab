"""Diagnostic: per-kernel durations and the idle gaps between consecutive kernels of one bench step, from a rocprofv3 kernel trace.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/kt --output-format csv -- python3 $R/bench.py --rows 250000 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-kernels
                 python tools/kernel_gaps.py gpurun_out/kt"""
import glob, sys
import pandas as pd
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
df = pd.concat([pd.read_csv(f) for f in fs]).sort_values("Start_Timestamp").reset_index(drop=True)
df["name"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("cd::", "")
df["dur"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
df["gap"] = (df["Start_Timestamp"] - df["End_Timestamp"].shift(1)) / 1e3
# the last full step: from the last prep16 / row_ratio kernel on
starts = df.index[df["name"].str.startswith("row_ratio")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2   # which step (index into the row_ratio launches; bench.py: W warm-ups, K timed steps, K fully bracketed ones)
if len(starts) >= 2:
    a, b = starts[which], starts[which + 1]
    step = df.iloc[a:b]
    print(f"one step: {len(step)} launches, wall {(step['End_Timestamp'].max() - step['Start_Timestamp'].min()) / 1e3:.1f} us, "
          f"kernel time {step['dur'].sum():.1f} us, gaps {step['gap'].iloc[1:].clip(lower=0).sum():.1f} us")
    for _, r in step.iterrows():
        print(f"  {r['name'][:60]:60s} dur {r['dur']:8.1f} us   gap before {r['gap']:7.1f} us")
