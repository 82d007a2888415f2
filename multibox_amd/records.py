"""Result records of detect.py:438-460 as JSON text.  numpy + json only (no torch): detect.py runs these functions in
worker PROCESSES, off the thread that feeds the GPU."""
from __future__ import annotations

import json

import numpy as np


def results_to_json_text(boxes, scores, count, image_ids):
    """One string per record of detect.py:438-443 ({"image_id", "bbox", "score"} in patch order), byte for byte what
    json.dumps() writes for the record dicts (default separators, floats by the same C encoder) -- without building 70
    dicts per patch: "[" + ", ".join(all records) + "]" == json.dumps(list of dicts)."""
    boxes, scores, count = np.asarray(boxes), np.asarray(scores), np.asarray(count)
    out = []
    for b in range(boxes.shape[0]):
        n = int(count[b])
        if n == 0:
            continue
        head = '{"image_id": %s, "bbox": [' % json.dumps(image_ids[b])
        bb = json.dumps(boxes[b, :n].tolist())[2:-2].split("], [")        # '[[a, b, c, d], [..]]' -> ['a, b, c, d', ..]
        sc = json.dumps(scores[b, :n].tolist())[1:-1].split(", ")
        out.extend([head + x + '], "score": ' + s + "}" for x, s in zip(bb, sc)])
    return out


def batch_chunk(boxes, scores, count, image_ids):
    """(number of records, the records of one batch joined with ", ") -- one picklable string per batch."""
    recs = results_to_json_text(boxes, scores, count, image_ids)
    return len(recs), ", ".join(recs)


def records_to_json(records):
    """The file detect.py:458-460 writes (json.dump of the record list) from record dicts, record strings or per-batch
    chunks of records (strings joined with ", "; empty chunks are skipped)."""
    if records and isinstance(records[0], str):
        return "[" + ", ".join(r for r in records if r) + "]"
    return json.dumps(records)
