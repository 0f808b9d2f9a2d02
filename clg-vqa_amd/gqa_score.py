"""Accuracy of a ``*_result.json`` written by ``EvaluatingModel`` against a GQA ground-truth file -- the metric of
volta/scripts/GQA_score.py:6-33: predictions whose questionId is missing from the truth file are skipped, the rest
score 1 when ``prediction == truth[questionId]["answer"]``.

    python -m clg_vqa_amd.gqa_score --preds_file val_result.json --truth_file testdev_balanced_questions.json
"""
import argparse
import json


def evaluate(preds_list, truth_dict):
    score, count = 0.0, 0
    for entry in preds_list:
        truth = truth_dict.get(entry["questionId"])
        if truth is None or "answer" not in truth:
            continue
        score += 1.0 if entry["prediction"] == truth["answer"] else 0.0
        count += 1
    if count == 0:
        raise ValueError("clg_vqa_amd.gqa_score: no prediction has a ground-truth entry")
    return score / count


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--preds_file", required=True)
    p.add_argument("--truth_file", required=True)
    a = p.parse_args(argv)
    s = 100 * evaluate(json.load(open(a.preds_file)), json.load(open(a.truth_file)))
    print(s)
    return s


if __name__ == "__main__":
    main()
