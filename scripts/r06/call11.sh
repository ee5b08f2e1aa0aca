#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_wgrad3_pipe.py tests/test_generator_module.py tests/test_discriminator_criterions.py tests/test_train_step.py -m gpu -q 2>&1 | tail -5 | cut -c1-300
timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "generator_256_vs_oracle or discriminator" 2>&1 | grep -E "parity-256|passed|failed" | cut -c1-420
