"""cProfile of a full run_lemon run on CIFAR-100-shaped synthetic data (diagnostic)."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemon_amd.run_lemon import main
n = sys.argv[1] if len(sys.argv) > 1 else "50000"
pr = cProfile.Profile(); pr.enable()
main(["--output_dir", "/tmp/lemon_e2e_prof", "--dataset", "cifar100", "--noise_type", "asymmetric", "--data_root", f"synthetic:{n}",
      "--clip_path", "random", "--knn_k", "50", "--encoder_batch", "1000"])
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
