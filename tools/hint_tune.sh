# sum of the per-iteration times 1..8 of tools/iter_trace.py (and of the listed points) for hint parameters (GPU box)
for rep in 1 2; do for wc in "2 1.5" "1 1.2" "1 1.5" "2 1.2"; do set -- $wc
  echo -n "W=$1 C=$2: "; SPKM_HINT_W=$1 SPKM_HINT_C=$2 python tools/iter_trace.py 2>/dev/null | awk '/^iter [1-8]:/{s+=$3; l+=$12} END{print s " ms over iterations 1..8, listed " l}'
done; done
