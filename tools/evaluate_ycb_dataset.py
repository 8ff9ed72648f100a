"""examples/evaluate_ycb_dataset.cpp over the device context (region + depth modality with measured occlusions,
single-region models, tracking from the first keyframe's ground truth):

    python tools/evaluate_ycb_dataset.py YCB_VIDEO_DIR EXTERNAL_DIR [sequence_id ...]

EXTERNAL_DIR holds poses/ground_truth/<sequence>_<body>.txt (the reference ships them as data/ycb-video_poses.zip)
and receives models/.  Prints ADD / ADD-S AUC and the mean step time per (sequence, body) and overall."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dobjecttracking_amd")

BODY_NAMES = ["002_master_chef_can", "003_cracker_box", "004_sugar_box", "005_tomato_soup_can", "006_mustard_bottle",
              "007_tuna_fish_can", "008_pudding_box", "009_gelatin_box", "010_potted_meat_can", "011_banana",
              "019_pitcher_base", "021_bleach_cleanser", "024_bowl", "025_mug", "035_power_drill", "036_wood_block",
              "037_scissors", "040_large_marker", "051_large_clamp", "052_extra_large_clamp", "061_foam_brick"]


def report(title, result):
    print("%s: execution_time = %g us, add auc = %g, adds auc = %g" %
          (title, result["complete_cycle"], result["add_auc"], result["adds_auc"]))


# one process per GPU (torch.distributed.run or any launcher that sets these): every process takes its share of runs
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))

if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit("usage: evaluate_ycb_dataset.py YCB_VIDEO_DIR EXTERNAL_DIR [sequence_id ...]")
    sequence_ids = [int(x) for x in sys.argv[3:]] or list(range(48, 60))  # evaluate_ycb_dataset.cpp:13
    _, overall = pkg.evaluation.evaluate_ycb_dataset(lambda: pkg.open_context(local_rank), sys.argv[1], sys.argv[2], sequence_ids,
                                                     BODY_NAMES, report=report, shard=(rank, world))
    report("all sequences, all bodies", overall)
