"""Host-side id hashing / string helpers of M-BEIR (drop-in for the parts of UniIR
src/data/preprocessing/utils.py that sit on the train / embed / retrieve path: :7-31 tables, :46-74 id hashing,
:77-107 lookups, :110-116 format_string, :198-269 jsonl helpers).  Pure Python; pinned by tests/golden/g9_host.json."""
import json

DATASET_IDS = {"VisualNews": 0, "Fashion200K": 1, "WebQA": 2, "EDIS": 3, "NIGHTS": 4, "OVEN": 5, "INFOSEEK": 6,
               "FashionIQ": 7, "CIRR": 8, "MSCOCO": 9}

_MODS = ["text", "image", "image,text"]
# task id = 3 * index(query modality) + index(candidate modality) with candidate order (image, text, image,text);
# spelled out because one entry ("image -> text,image") is written in the other order upstream
MBEIR_TASK = {
    "text -> image": 0, "text -> text": 1, "text -> image,text": 2,
    "image -> text": 3, "image -> image": 4, "image -> text,image": 5,
    "image,text -> text": 6, "image,text -> image": 7, "image,text -> image,text": 8,
}

DATASET_CAN_NUM_UPPER_BOUND = 10_000_000   # candidates per dataset
DATASET_QUERY_NUM_UPPER_BOUND = 500_000    # queries per dataset


def _split(xid):
    ds, n = xid.split(":")
    return int(ds), int(n)


def hash_qid(qid):
    ds, n = _split(qid)
    return ds * DATASET_QUERY_NUM_UPPER_BOUND + n


def unhash_qid(hashed_qid):
    return f"{hashed_qid // DATASET_QUERY_NUM_UPPER_BOUND}:{hashed_qid % DATASET_QUERY_NUM_UPPER_BOUND}"


def hash_did(did):
    ds, n = _split(did)
    return ds * DATASET_CAN_NUM_UPPER_BOUND + n


def unhash_did(hashed_did):
    return f"{hashed_did // DATASET_CAN_NUM_UPPER_BOUND}:{hashed_did % DATASET_CAN_NUM_UPPER_BOUND}"


def get_dataset_id(dataset_name):
    return DATASET_IDS.get(dataset_name)


def get_dataset_name(xid):
    ds = int(xid.split(":")[0])
    return next((name for name, i in DATASET_IDS.items() if i == ds), None)


def get_mbeir_task_id(source_modality, target_modality):
    return MBEIR_TASK.get(f"{source_modality} -> {target_modality}")


def get_mbeir_task_name(task_id):
    return next((name for name, i in MBEIR_TASK.items() if i == task_id), None)


def get_mbeir_query_modality_cand_modality_from_task_id(task_id):
    name = get_mbeir_task_name(task_id)
    return name.split(" -> ") if name is not None else None


def format_string(s):
    """strip, drop carriage returns and surrounding double quotes, capitalise, make sure it ends a sentence"""
    s = (s or "").replace("\r", "").strip().strip('"')
    if not s:
        return s
    s = s[0].upper() + s[1:]
    return s if s[-1] in ".?!" else s + "."


def load_jsonl_as_list(path):
    with open(path, "r") as f:
        return [json.loads(line) for line in f if line.strip()]


def save_list_as_jsonl(data, filename, mode="w"):
    """(data, filename) like the reference (src/data/preprocessing/utils.py:198)"""
    with open(filename, mode) as f:
        for e in data:
            f.write(json.dumps(e) + "\n")


def count_entries_in_file(filename):
    """-> (number of entries, entries) of a .jsonl / .json file (reference :257-269)"""
    if filename.endswith(".jsonl"):
        data = load_jsonl_as_list(filename)
    elif filename.endswith(".json"):
        with open(filename, "r") as f:
            data = json.load(f)
    else:
        raise ValueError("Unsupported file format. Only .json and .jsonl are supported.")
    return len(data), data


def print_mbeir_format_dataset_stats(data, cand_pool_dict):
    """console summary of an M-BEIR instance file after hard-negative mining (reference :548-563 prints a longer
    per-modality table; the numbers a training run needs to sanity-check are kept): instances, candidate list lengths,
    modality pairs of the positives"""
    n = max(1, len(data))
    pairs = {}
    for e in data:
        for did in e.get("pos_cand_list", []):
            cand = cand_pool_dict.get(did)
            key = f"{e.get('query_modality')} -> {cand.get('modality') if cand else '?'}"
            pairs[key] = pairs.get(key, 0) + 1
    print(f"--- INSTANCES ---\n\t{len(data)}")
    print(f"--- AVG_POS_CAND ---\n\t{sum(len(e.get('pos_cand_list', [])) for e in data) / n:.1f}")
    print(f"--- AVG_NEG_CAND ---\n\t{sum(len(e.get('neg_cand_list', [])) for e in data) / n:.1f}")
    print("--- QUERY -> POSITIVE MODALITY ---")
    for key in sorted(pairs):
        print(f"\t{key}: {pairs[key]}")


def load_mbeir_format_pool_file_as_dict(pool_file_path, doc_key_to_content=False, key_type="did"):
    pool = {}
    for e in load_jsonl_as_list(pool_file_path):
        if doc_key_to_content:
            key = e["did"] if key_type == "did" else f"{e.get('txt') or ''}{e.get('img_path') or ''}"
            pool[key] = e
        else:
            pool.setdefault(f"{e.get('txt') or ''}{e.get('img_path') or ''}", []).append(e["did"])
    return pool
