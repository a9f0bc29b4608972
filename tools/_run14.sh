mkdir -p gpurun_out/r6m
for b in 16 32; do
echo "== batch $b, this tree (round 6)" >> gpurun_out/r6m/cosine_b.txt
CK_BATCH=$b python tools/compact_keys_cosine.py 2>&1 | grep "seed\|Error" >> gpurun_out/r6m/cosine_b.txt
echo "== batch $b, round-5 tree (_other: 757f73b)" >> gpurun_out/r6m/cosine_b.txt
(cd _other && CK_BATCH=$b DL_TREE=$PWD python ../tools/compact_keys_cosine.py 2>&1 | grep "seed\|Error") >> gpurun_out/r6m/cosine_b.txt
done
cat gpurun_out/r6m/cosine_b.txt
