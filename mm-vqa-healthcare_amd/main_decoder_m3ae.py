"""Entry point with the reference's CLI (`python main_decoder_m3ae.py with k=v ... named_config ...`,
run_scripts/finetune_m3ae_decoder.sh): reference main_decoder_m3ae.py without sacred / Lightning -- see
m3ae_amd/trainer.py and m3ae_amd/modules/m3ae_decoder.py (8f-3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from m3ae_amd.trainer import run  # noqa: E402

if __name__ == "__main__":
    run(sys.argv[1:], head="decoder")
