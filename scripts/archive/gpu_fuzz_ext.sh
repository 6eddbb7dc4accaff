#!/bin/bash
# one-off bug hunt: the randomized differential test with fresh seeds (HIPADJ_FUZZ_BASE shifts the seed range)
cd ${GRAFT_REPO_ROOT:-.}
for base in ${@:-1000 6000}; do
  HIPADJ_FUZZ_BASE=$base timeout 900 python -m pytest tests -m gpu -q -k "randomized_configurations_match" 2>&1 | grep -v "^  File" | tail -12
done
