# sum of the per-iteration times 1..8 of tools/iter_trace.py for hint parameters (GPU box)
for w in 1 2 4; do for c in 1.2 1.5 2.0; do
  echo -n "W=$w C=$c: "; SPKM_HINT_W=$w SPKM_HINT_C=$c python tools/iter_trace.py 2>/dev/null | awk '/^iter [1-8]:/{s+=$3} /^iter 1:/{l=$NF" "$(NF-2)} END{print s " ms over iterations 1..8"}'
done; done
