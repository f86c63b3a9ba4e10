"""`python -m sisr_cli train --parameters cfg.toml` / `python -m sisr_cli eval --config cfg.toml`
(the reference's train_sisr / eval_sisr console scripts, Code/setup.py:13-22)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sisr_amd  # noqa: E402

if __name__ == "__main__":
    sisr_amd.cli.main()
