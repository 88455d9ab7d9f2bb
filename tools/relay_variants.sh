# A/B of the leader's hand-over to the other workgroups of an armed launch: OTTI_RELAY=1 (default): its copies (eight, 4 KiB apart) carry values + number +
# tag and are polled as one batch of loads; OTTI_RELAY=0: one copy, the number is polled alone and the values are loaded after it.
# (profiles/r4_relay_variants.txt also holds the run with round mails as seven unfenced stores, OTTI_TAIL_MAIL_FENCE=0 at the time: 13 us late — that form
# is gone; the tail now mails its line in one store instruction, tools/pollprobe.hip.)
for v in "OTTI_RELAY=1" "OTTI_RELAY=0" "OTTI_RELAY=1" "OTTI_RELAY=0"; do
  echo "=== $v"
  env $v OTTI_TRACE=1 python tools/snark_probe.py 20 4 2>&1 | grep "  prove\|pcbatch\|rror" | tail -3 | cut -c1-330
  env $v python bench.py --no-snark --no-sweep --no-e2e --in-flight -1 --no-cpu-baseline 2>/dev/null | python tools/benchsum.py | head -1
done
