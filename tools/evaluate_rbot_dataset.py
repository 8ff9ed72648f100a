"""examples/evaluate_rbot_dataset.cpp over the device context (region modality, sequences without modelled
occlusions):

    python tools/evaluate_rbot_dataset.py RBOT_DATASET_DIR EXTERNAL_DIR [body ...]

Prints the success rate and the mean step time per (sequence, body) and overall, like
RBOTEvaluator::VisualizeFinalResult."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dobjecttracking_amd")


def report(title, result):
    print("-" * 80)
    print("%s:\nsuccess rate = %g\ncomplete cycle = %g us" % (title, result["tracking_success"], result["complete_cycle"]))


# one process per GPU (torch.distributed.run or any launcher that sets these): every process takes its share of runs
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))

if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit("usage: evaluate_rbot_dataset.py RBOT_DATASET_DIR EXTERNAL_DIR [body ...]")
    ev = pkg.evaluation
    bodies = sys.argv[3:] or ev.RBOT_BODY_NAMES
    _, overall = ev.evaluate_rbot_dataset(lambda: pkg.open_context(local_rank), sys.argv[1], sys.argv[2], bodies, report=report, shard=(rank, world))
    report("all_sequences_all_bodies", overall)
