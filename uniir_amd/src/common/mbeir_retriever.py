"""Index + brute-force retrieval + Recall@k for M-BEIR on MI355X (drop-in for UniIR src/common/mbeir_retriever.py).

create_index (:34-129), search_index (:188-232), compute_recall_at_k (:149-166), run_retrieval (:312-603) and the CLI
(:711-761) keep their names, arguments, file names and outputs.  FAISS ("IDMap,Flat", inner product, sharded over the
visible GPUs) is replaced by uniir_amd.retrieval (libuniir_hip.so): the pool stays in its stored fp16 form with fp32
inverse norms, is sharded row-wise over the visible GPUs, each shard returns its exact top-k, and a k-way merge on
(score desc, id asc) gives the result -- identical to the unsharded search.  The ".index" file is our own container
(np.savez: emb fp16, ids int64), only ever read back by search_index, like the reference's FAISS file.
Not copied: the reference re-reads and re-uploads the index on every search_index call (:197-206); here device shards
are cached per index file.  The reference's main() passes an unbound query_embedder_config to run_retrieval (:757);
here it defaults to None.
"""
import os as _os
import sys as _sys

_SRC = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))          # <repo>/uniir_amd/src
for _p in (_os.path.dirname(_os.path.dirname(_SRC)), _SRC, _os.path.join(_SRC, "common")):
    if _p not in _sys.path:
        _sys.path.insert(0, _p)

import argparse
import csv
import gc
import os
from collections import defaultdict
from datetime import datetime

import numpy as np
import torch

from config import OmegaConf
from data.preprocessing.utils import (count_entries_in_file, get_mbeir_task_name, load_jsonl_as_list,
                                      load_mbeir_format_pool_file_as_dict, print_mbeir_format_dataset_stats,
                                      save_list_as_jsonl, unhash_did, unhash_qid)

_SHARD_CACHE = {}


def create_index(config):
    index_config = config.index_config
    expt = config.experiment.path_suffix
    pools = index_config.cand_pools_config
    assert pools.enable_idx, "Indexing is not enabled for candidate pool"
    split = "cand_pool"
    for name in pools.cand_pools_name_to_idx:
        name = name.lower()
        embed_dir = os.path.join(config.uniir_dir, index_config.embed_dir_name, expt, split)
        emb = np.load(os.path.join(embed_dir, f"mbeir_{name}_{split}_embed.npy"))
        ids = np.load(os.path.join(embed_dir, f"mbeir_{name}_{split}_ids.npy")).astype(np.int64)
        assert len(ids) == len(set(ids.tolist())), "IDs should be unique"
        assert index_config.faiss_config.dim == emb.shape[1], \
            "The dimension of the index does not match the dimension of the embeddings!"
        index_path = os.path.join(config.uniir_dir, index_config.index_dir_name, expt, split, f"mbeir_{name}_{split}.index")
        os.makedirs(os.path.dirname(index_path), exist_ok=True)
        with open(index_path, "wb") as f:   # stored un-normalised fp16 (exact); normalisation happens on the device
            np.savez(f, emb=emb.astype(np.float16), ids=ids)
        print(f"Successfully indexed {len(ids)} documents\nIndex saved to: {index_path}")
        del emb, ids
        gc.collect()


def compute_recall_at_k(relevant_docs, retrieved_indices, k):
    """hit rate: 1.0 when any relevant doc is among the first k retrieved (the CLIP / BLIP convention)"""
    if not relevant_docs:
        return 0.0
    return 1.0 if set(relevant_docs) & set(retrieved_indices[:k]) else 0.0


def _device_shards(cand_index_path):
    """row-shard the pool over the visible GPUs (FAISS shard=True semantics); cached per (path, mtime)"""
    from uniir_amd import retrieval
    key = (cand_index_path, os.path.getmtime(cand_index_path))
    if key not in _SHARD_CACHE:
        _SHARD_CACHE.clear()
        with np.load(cand_index_path) as z:
            emb, ids = z["emb"], z["ids"]
        ndev = max(1, torch.cuda.device_count())
        # UNIIR_RETRIEVER_SHARDS (tests): that many row shards, placed round-robin on the visible devices -- the N > 1 shard loop
        # and the merge on a box with one GPU
        ngpu = int(os.environ.get("UNIIR_RETRIEVER_SHARDS", "0")) or ndev
        per = -(-len(ids) // ngpu)
        shards = []
        for g in range(ngpu):
            lo, hi = min(g * per, len(ids)), min((g + 1) * per, len(ids))
            if hi <= lo:
                continue
            dev = torch.device("cuda", g % ndev)
            with torch.cuda.device(dev):
                shards.append(retrieval.PoolShard(torch.from_numpy(emb[lo:hi]).to(dev), torch.from_numpy(ids[lo:hi]).to(dev)))
        print(f"Retriever: {len(ids)} documents in {len(shards)} shard(s) over {min(ngpu, ndev)} GPU(s)")
        _SHARD_CACHE[key] = shards
    return _SHARD_CACHE[key]


def search_index_with_batch(query_embeddings_batch, shards, num_cand_to_retrieve=10):
    from uniir_amd import retrieval
    outs = []
    for sh in shards:                      # launches are asynchronous per device: the shards search concurrently
        dev = sh.emb.device
        with torch.cuda.device(dev):
            q = torch.from_numpy(query_embeddings_batch).to(dev)
            outs.append(retrieval.search_shard(sh, q, num_cand_to_retrieve))
    dev0 = shards[0].emb.device
    with torch.cuda.device(dev0):
        s = torch.stack([o[0].to(dev0) for o in outs])
        i = torch.stack([o[1].to(dev0) for o in outs])
        ms, mi = (s[0], i[0]) if len(outs) == 1 else retrieval.merge_shards(s, i)
        return ms.cpu().numpy(), mi.cpu().numpy()


def search_index(query_embed_path, cand_index_path, batch_size=10, num_cand_to_retrieve=10):
    queries = np.load(query_embed_path).astype(np.float16)   # stored fp16; normalised on the device in fp32
    print(f"Retriever: loaded query embeddings from {query_embed_path} with shape: {queries.shape}")
    shards = _device_shards(cand_index_path)
    dists, idxs = [], []
    for i in range(0, len(queries), max(1, batch_size)):
        d, ix = search_index_with_batch(queries[i:i + batch_size], shards, num_cand_to_retrieve)
        dists.append(d)
        idxs.append(ix)
    return np.vstack(dists), np.vstack(idxs)


_DATASET_ORDER = ["visualnews_task0", "mscoco_task0", "fashion200k_task0", "webqa_task1", "edis_task2", "webqa_task2",
                  "visualnews_task3", "mscoco_task3", "fashion200k_task3", "nights_task4", "oven_task6", "infoseek_task6",
                  "fashioniq_task7", "cirr_task7", "oven_task8", "infoseek_task8"]
_RECALLS = ["Recall@1", "Recall@5", "Recall@10", "Recall@20", "Recall@50"]


def run_file_line(qid, doc_id, rank, score, run_id, task_id):
    """one TREC-style hit: query-id Q0 document-id rank score run-id task-id (reference mbeir_retriever.py:438-443; pinned by
    tests/golden/g9_host.json["runfile"]).  `score` is the numpy float32 the search returned: it prints with float32 repr digits."""
    return f"{qid} Q0 {doc_id} {rank} {score} {run_id} {task_id}\n"


def run_retrieval(config, query_embedder_config=None):
    rc = config.retrieval_config
    expt = config.experiment.path_suffix
    results_dir = os.path.join(config.uniir_dir, rc.results_dir_name, expt)
    run_dir, tsv_dir = os.path.join(results_dir, "run_files"), os.path.join(results_dir, "final_tsv")
    for d in (run_dir, os.path.join(results_dir, "retrieved_candidates"), tsv_dir):
        os.makedirs(d, exist_ok=True)
    if rc.get("raw_retrieval"):
        raise NotImplementedError("raw_retrieval (UniRAG candidate dump) is outside the MI355X hot path, see DESIGN.md")
    index_dir = os.path.join(config.uniir_dir, rc.index_dir_name, expt, "cand_pool")
    qrel_dir = os.path.join(config.mbeir_data_dir, rc.qrel_dir_name)
    results = []
    for split in ("train", "val", "test"):
        dc = rc.get(f"{split}_datasets_config")
        if not (dc and dc.enable_retrieve):
            continue
        names, pools = list(dc.datasets_name), list(dc.correspond_cand_pools_name)
        qrels, metrics = list(dc.correspond_qrels_name), list(dc.correspond_metrics_name)
        assert len(names) == len(pools) == len(qrels) == len(metrics), "Mismatch between datasets and candidate pools and qrels."
        embed_dir = os.path.join(config.uniir_dir, rc.embed_dir_name, expt, split)
        from utils import load_qrel
        for dataset, pool, qrel_name, metric_names in zip(names, pools, qrels, metrics):
            dataset, pool, qrel_name = dataset.lower(), pool.lower(), qrel_name.lower()
            print(f"\nRetriever: Retrieving for query:{dataset} | split:{split} | from cand_pool:{pool}")
            qrel, qid_to_task = load_qrel(os.path.join(qrel_dir, split, f"mbeir_{qrel_name}_{split}_qrels.txt"))
            qids = np.load(os.path.join(embed_dir, f"mbeir_{dataset}_{split}_ids.npy"))
            recalls = [m.strip() for m in metric_names.split(",") if "recall" in m.lower()]
            k = max(int(m.split("@")[1]) for m in recalls)
            dist_mat, idx_mat = search_index(os.path.join(embed_dir, f"mbeir_{dataset}_{split}_embed.npy"),
                                             os.path.join(index_dir, f"mbeir_{pool}_cand_pool.index"),
                                             batch_size=qids.shape[0], num_cand_to_retrieve=k)
            run_id = f"mbeir_{dataset}_{'union' if pool == 'union' else 'single'}_pool_{split}_k{k}"
            by_task = defaultdict(lambda: defaultdict(list))
            with open(os.path.join(run_dir, f"{run_id}_run.txt"), "w") as rf:
                for qi, (ds, ix) in enumerate(zip(dist_mat, idx_mat)):
                    qid = unhash_qid(int(qids[qi]))
                    task = qid_to_task[qid]
                    docs = [unhash_did(int(h)) for h in ix]
                    for rank, (doc, score) in enumerate(zip(docs, ds), start=1):
                        rf.write(run_file_line(qid, doc, rank, score, run_id, task))
                    for m in recalls:
                        by_task[task][m].append(compute_recall_at_k(qrel[qid], docs, int(m.split("@")[1])))
            for task, vals in by_task.items():
                row = {"TaskID": int(task), "Task": get_mbeir_task_name(int(task)), "Dataset": dataset, "Split": split,
                       "CandPool": pool}
                for m in recalls:
                    row[m] = round(sum(vals[m]) / len(vals[m]), 4)
                    print(f"Retriever: Mean {m}: {row[m]}")
                results.append(row)
    order = {n: i + 1 for i, n in enumerate(_DATASET_ORDER)}
    results.sort(key=lambda r: (r["TaskID"], order.get(r["Dataset"].lower(), 99), {"val": 1, "test": 2}.get(r["Split"], 99),
                                99 if r["CandPool"] == "union" else 0))
    if rc.get("write_to_tsv"):
        grouped = defaultdict(dict)
        for r in results:
            grouped[(r["TaskID"], r["Task"], r["Dataset"], r["Split"])][r["CandPool"]] = {m: r.get(m) for m in _RECALLS}
        path = os.path.join(tsv_dir, f"eval_results_{datetime.now().strftime('%m-%d-%H')}.tsv")
        with open(path, "w", newline="") as f:
            w = csv.writer(f, delimiter="\t")
            w.writerow(["TaskID", "Task", "Dataset", "Split", "Metric", "CandPool", "Value", "UnionPool", "UnionValue"])
            for (tid, task, dataset, split), pools in grouped.items():
                union = pools.get("union", {})
                for m in _RECALLS:
                    for pool, vals in pools.items():
                        if pool == "union" or vals.get(m) is None:
                            continue
                        w.writerow([tid, task, dataset, split, m, pool, vals[m]] +
                                   (["union", union.get(m, "N/A")] if union else ["", ""]))
        print(f"Retriever: Results saved to {path}")
    return results


def select_hard_negatives(retrieved_dids, pos_cand_list, neg_cand_list, num_hard_negs):
    """retrieved candidates that are neither positives nor already-listed negatives, in rank order; a short non-empty list
    is repeated cyclically up to num_hard_negs, a long one cut there (reference :664-680)"""
    known = set(pos_cand_list) | set(neg_cand_list)
    hard = [d for d in retrieved_dids if d not in known]
    if not hard:
        print("Warning: hard_negatives list is empty.")
        return hard
    if len(hard) < num_hard_negs:
        hard = [hard[i % len(hard)] for i in range(num_hard_negs)]
    return hard[:num_hard_negs]


def run_hard_negative_mining(config):
    """src/common/mbeir_retriever.py:606-708: top-k of every train query of the first dataset over the first candidate
    pool (device brute force instead of FAISS), filtered into hard negatives and appended to neg_cand_list; output
    <mbeir_data_dir>/train/<hard_negs_dir_name>/mbeir_<dataset>_hard_negs_train.jsonl"""
    rc = config.retrieval_config
    expt = config.experiment.path_suffix
    tc = rc.train_datasets_config
    assert tc.enable_retrieve, "Hard negative mining is not enabled for training data"
    dataset_name, split = tc.datasets_name[0].lower(), "train"      # only the first dataset / pool, like the reference
    query_data_list = load_jsonl_as_list(os.path.join(config.mbeir_data_dir, "train", f"mbeir_{dataset_name}_{split}.jsonl"))
    embed_dir = os.path.join(config.uniir_dir, rc.embed_dir_name, expt, split)
    query_ids = np.load(os.path.join(embed_dir, f"mbeir_{dataset_name}_{split}_ids.npy"))
    embed_path = os.path.join(embed_dir, f"mbeir_{dataset_name}_{split}_embed.npy")
    pool_name, pool_split = tc.correspond_cand_pools_name[0].lower(), "cand_pool"
    index_path = os.path.join(config.uniir_dir, rc.index_dir_name, expt, pool_split, f"mbeir_{pool_name}_{pool_split}.index")
    print("-" * 30)
    print(f"Hard Negative mining, Datasets: {dataset_name}, Candidate Pools: {pool_name}")
    print("-" * 30)
    # the reference searches all queries in one FAISS call; here QUERY_CHUNK-sized sweeps over the resident shards
    _, retrieved = search_index(embed_path, index_path, batch_size=max(1, query_ids.shape[0]), num_cand_to_retrieve=rc.k)
    assert len(query_ids) == len(retrieved)
    for i, hashed_qid in enumerate(query_ids):
        entry = query_data_list[i]
        assert unhash_qid(hashed_qid) == entry["qid"]
        dids = [unhash_did(h) for h in retrieved[i]]
        entry["neg_cand_list"].extend(select_hard_negatives(dids, entry["pos_cand_list"], entry["neg_cand_list"],
                                                            rc.num_hard_negs))
    out_path = os.path.join(config.mbeir_data_dir, "train", rc.hard_negs_dir_name,
                            f"mbeir_{dataset_name}_hard_negs_{split}.jsonl")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    save_list_as_jsonl(query_data_list, out_path)
    total, data = count_entries_in_file(out_path)
    print(f"MBEIR Train Data with Hard Negatives saved to {out_path}")
    print(f"Total number of entries in {out_path}: {total}")
    pool_path = os.path.join(config.mbeir_data_dir, pool_split, f"mbeir_{pool_name}_{pool_split}.jsonl")
    print_mbeir_format_dataset_stats(data, load_mbeir_format_pool_file_as_dict(pool_path, doc_key_to_content=True, key_type="did"))


def parse_arguments():
    p = argparse.ArgumentParser(description="MI355X brute-force retrieval pipeline")
    p.add_argument("--uniir_dir", type=str, default="/data/UniIR")
    p.add_argument("--mbeir_data_dir", type=str, default="/data/UniIR/mbeir_data")
    p.add_argument("--config_path", default="config.yaml")
    p.add_argument("--query_embedder_config_path", default="")
    p.add_argument("--enable_create_index", action="store_true")
    p.add_argument("--enable_hard_negative_mining", action="store_true")
    p.add_argument("--enable_retrieval", action="store_true")
    return p.parse_args()


def main():
    args = parse_arguments()
    config = OmegaConf.load(args.config_path)
    config.uniir_dir, config.mbeir_data_dir = args.uniir_dir, args.mbeir_data_dir
    print(OmegaConf.to_yaml(config, sort_keys=False))
    if args.enable_hard_negative_mining:
        run_hard_negative_mining(config)
    if args.enable_create_index:
        create_index(config)
    if args.enable_retrieval:
        run_retrieval(config, None)


if __name__ == "__main__":
    main()
