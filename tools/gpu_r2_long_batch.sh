set -e
cd /tmp && rm -rf lb && mkdir lb && cd lb
R=$GRAFT_REPO_ROOT
echo "seed=5 -o ours.wav -r 16000 -d 600 -j 1 -s 5.76 -n 20 | -v 2 -g 3" > m.txt
VS_WAV_HEADER=72 $R/voice_synth_amd/bin/vs_batch m.txt
VS_SEED=5 VS_DRAWLOG=/tmp/lb/dl $R/oracle/_ref/flowgen_shimmer -o g.wav -r 16000 -d 600 -j 1 -s 5.76 -n 20 > /dev/null
VS_SEED=5 VS_DRAWLOG=/tmp/lb/dl $R/oracle/_ref/vowel -i g.wav -o ref.wav -v 2 -g 3 > /dev/null
ls -l ours.wav ref.wav; cmp ours.wav ref.wav && echo "10-minute utterance: vs_batch output equals the reference's, byte for byte"
# the two drop-in programs on the same 10-minute utterance: files and stdout against the reference's
mkdir -p ours ref
(cd ours && VS_SEED=5 VS_WAV_HEADER=72 $R/voice_synth_amd/bin/flowgen_shimmer -o g.wav -r 16000 -d 600 -j 1 -s 5.76 -n 20 > fg.txt && VS_SEED=5 VS_WAV_HEADER=72 $R/voice_synth_amd/bin/vowel -i g.wav -o o.wav -v 2 -g 3 > vw.txt)
(cd ref && VS_SEED=5 VS_DRAWLOG=/tmp/lb/dl $R/oracle/_ref/flowgen_shimmer -o g.wav -r 16000 -d 600 -j 1 -s 5.76 -n 20 > fg.txt && VS_SEED=5 VS_DRAWLOG=/tmp/lb/dl $R/oracle/_ref/vowel -i g.wav -o o.wav -v 2 -g 3 > vw.txt)
cmp ours/g.wav ref/g.wav && cmp ours/o.wav ref/o.wav && cmp ours/fg.txt ref/fg.txt && cmp ours/vw.txt ref/vw.txt && echo "10-minute utterance: both programs equal the reference's files and stdout ($(wc -l < ref/fg.txt) stdout lines)"
