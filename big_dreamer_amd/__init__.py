"""MI355X-native hot path of jgsimard/big-dreamer (Dreamer.train_step) -- see DESIGN.md."""
import os as _os

# The engine drives six HIP streams and a data-parallel run adds four RCCL communicators with a stream each, plus the
# caller's.  The HIP runtime multiplexes the streams of a priority level onto GPU_MAX_HW_QUEUES hardware queues (default 4),
# and commands of streams that share a queue execute in the order of submission: a gradient all-reduce then waits for
# whatever the queue's other stream was given before it.  One-rank RCCL rehearsal (tools/r03_dp_rehearsal.sh,
# tools/dp_rehearsal.py): with 4, 8 or 12 queues 14.7 instead of 12.8 ms/step at configs[2] and 3.45 instead of 2.86 at
# configs[1] from a library caller; with 16 or more every case runs at the rate of the run without collectives.  The knob is a
# process-level runtime setting read when HIP initialises, so it is set here, at import, unless the user has chosen a value
# (a process that has already initialised HIP keeps the runtime's default: set it in the environment then).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
