#!/bin/bash
# A/B of builds of the library on the full evaluation shape: scripts/exp_variants.sh <model> <lib> [<lib> ...]
# (exp_eval.py --lib loads another build; the variants are built by hand from csrc/*.o and a differently compiled eval.o)
model=$1; shift
for lib in "$@"; do
  echo "== $lib $model"
  timeout -k 10 150 python scripts/exp_eval.py --users 1000000 --items 200000 --clicks 100000000 --model $model --reps 3 --lib $lib || exit 1
done
