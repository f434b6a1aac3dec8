"""Golden vectors for the tokeniser from the REFERENCE's own code (run in the build container, where /root/reference exists).

util/text_processing.py:9-67 (load_vocab_dict_from_file, sentence2vocab_indices, preprocess_sentence, preprocess_sentence_lstm) uses `re`
alone, but the module calls nltk.download(...) at import time (:6-7) and nltk is not installed: only those two import-time calls are
neutralised (an empty `nltk` module object whose download() does nothing); every function under test is the reference's, unmodified.
Inputs: expressions from the reference's data/referit_query_test.json and a few edge cases; vocabularies data/vocabulary_Gref.txt and
data/vocabulary_referit.txt.  Output: tests/golden/ref_text_processing.json (sentences + ids + lengths: data only).

    python tests/golden/make_text_fixtures.py
"""
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    stub = types.ModuleType("nltk")
    stub.download = lambda *a, **k: None
    sys.modules.setdefault("nltk", stub)
    spec = importlib.util.spec_from_file_location("ref_text_processing", os.path.join(REF, "util/text_processing.py"))
    tp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tp)
    q = json.load(open(os.path.join(REF, "data/referit_query_test.json")))
    sents = []
    for k in sorted(q)[::97][:160]:
        sents.extend(q[k][:1])
    sents += ["The man.", "a  dog , running; fast!", "left-most zebra's head (behind the tree).", "UPPER left", "x",
              " ".join(["word"] * 30), "qwertyuiopasdf unknownword", "person . ", "two words."]
    out = {"sentences": sents, "T": [20, 8], "vocab": {}}
    for vname in ("vocabulary_Gref.txt", "vocabulary_referit.txt"):
        vd = tp.load_vocab_dict_from_file(os.path.join(REF, "data", vname))
        # the slice of the vocabulary these sentences touch (word -> line number), so that the test needs no file from /root/reference
        toks = {w.lower() for s in sents for w in tp.SENTENCE_SPLIT_REGEX.split(s.strip()) if len(w.strip()) > 0}
        res = {"size": len(vd), "pad": vd["<pad>"], "unk": vd["<unk>"], "lstm": {}, "front": {}, "raw": [tp.sentence2vocab_indices(s, vd) for s in sents],
               "subset": {w: vd[w] for w in sorted(toks) if w in vd}}
        for T in out["T"]:
            pairs = [tp.preprocess_sentence_lstm(s, vd, T) for s in sents]
            res["lstm"][str(T)] = {"ids": [p[0] for p in pairs], "len": [p[1] for p in pairs]}
            res["front"][str(T)] = [tp.preprocess_sentence(s, vd, T) for s in sents]
        out["vocab"][vname] = res
    # the vocabulary files themselves are reference DATA the product also reads: their first / last lines pin the file identity
    with open(os.path.join(HERE, "ref_text_processing.json"), "w") as f:
        json.dump(out, f)
    print("wrote", len(sents), "sentences")


if __name__ == "__main__":
    main()
