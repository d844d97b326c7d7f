"""RecommenderALS - the drop-in facade for the implicit-ALS path of reclibwh.

The reference documents `reclibwh.recommender.recommenderALS.RecommenderALS` (README.md:71,138)
and uses it in notebooks/ego_graph_viz.ipynb (cells 7, 9, 15), but the class itself is not in
the repository.  Its surface is therefore taken from the sibling RecommenderMF usage block
(README.md:26-42: constructor(mode, n_users, n_items, <algo>_kwargs, model_path),
input_data(d), train(), save(path)) and from the notebook (mode='predict', predict(u, i),
similar_to([item])); numerics follow reclibwh/core/ALS.py (ALSRef / ALSTrainer).

Saved layout (ALSRef.save, ALS.py:143-149): <path>/X.npy and <path>/Y.npy, float64 row-major
[n_users, f] and [n_items, f] (the P / Q factors), plus algorithm_config.json
({epochs, start_epoch, alpha, lamb}, ALS.py:264-283) and eval.csv (epoch,score;
EvalProto.py:93-94) when evaluation ran.
"""
import json
import os

import numpy as np
import torch

from . import metrics, ops
from .als import ALSEngine, init_factors

DEFAULTS = dict(f=20,           # ALS.json.template:26 (the reference's layer width)
                lamb=1e-06,     # ALS.py:42, ALS.py:264
                alpha=40.0,     # ALS.py:42, ALS.py:264
                epochs=10,      # ALSRef steps (ALS.py:42)
                init="ref",     # "ref": ALS.py:59-65, "tf": ALS.py:255-257
                seed=42,
                phantom_rows=0,     # >0: TensorFlow-path layout, layers get extra never-updated rows
                                    # that still enter the Gram (ALS.py:243, ALS.py:456)
                chunk_nnz=None,     # entries per chunk item of a long row; None = ops.default_chunk_nnz(f)
                solver="auto",      # "cholesky" (exact, f <= 128), "cg", or "auto" = exact where it exists
                cg_steps=None,      # conjugate-gradient steps per row and half-sweep when solver is "cg"; None =
                                    # als.default_cg_steps(f) = max(10, ceil(f / 6)), the count at which 15 iterations
                                    # give the exact solve's MAP@200 / precision@10 / RMSE to 3 decimals
                short_rows="auto",  # exact solver: rows with fewer ratings than factors solved in their entry-space form
                                    # ("auto": where it pays, als.short_rows_pay; True / False)
                med_score=3.0,      # ALS.py:402
                eval_lim=200)       # EvalProto.py:22


class RecommenderALS:
    def __init__(self, mode="train", n_users=None, n_items=None, als_kwargs=None, model_path=None, device=None):
        if mode not in ("train", "predict"):
            raise ValueError("mode must be 'train' or 'predict'")
        self.mode = mode
        self.model_path = model_path
        kw = dict(DEFAULTS)
        for k, v in (als_kwargs or {}).items():
            k = {"K": "f", "steps": "epochs"}.get(k, k)
            if k not in kw:
                raise ValueError("unknown als_kwargs entry %r" % k)
            kw[k] = v
        self.config = kw
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("RecommenderALS needs a GPU: the HIP path has no CPU fallback")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.n_users, self.n_items = n_users, n_items
        self.X = self.Y = None
        self.Utrain = self.Utest = None
        self._eval_keys = None         # sorted (user, item) keys of the training pairs / relevant test pairs, on the device
        self.engine = None
        self._foldin_cache = None
        self._exclusions = None
        self._prepared = None          # the item factor in the score kernel's operand form, built on the first recommend()
        self.start_epoch = 0
        self.trace, self.eval_epochs, self.eval_scores = [], [], []
        if mode == "predict":
            if model_path is None:
                raise ValueError("mode='predict' needs model_path")
            self.load(model_path)

    # ------------------------------------------------------------------ data
    def input_data(self, data):
        """`data`: an object with make_training_datasets(dtype='sparse') -> (Utrain, Utest) and
        N / M properties (the ExplicitData contract, reclibwh/utils/ItemMetadata.py:378-392), a
        (Utrain, Utest) tuple of scipy sparse matrices, or one scipy sparse matrix [users, items]."""
        import scipy.sparse as sps
        if hasattr(data, "make_training_datasets"):
            Utrain, Utest = data.make_training_datasets(dtype="sparse")
        elif isinstance(data, (tuple, list)):
            Utrain, Utest = data
        else:
            Utrain, Utest = data, None
        Utrain = sps.csr_matrix(Utrain); Utrain.sort_indices()
        if Utrain.nnz and Utrain.data.min() < 0:
            raise ValueError("ratings must be >= 0 (implicit confidence c = 1 + alpha r)")
        self.Utrain = Utrain
        self._eval_keys = None
        self._exclusions = None
        self.Utest = None if Utest is None else sps.csr_matrix(Utest)
        if self.Utest is not None:
            self.Utest.sort_indices()
        N, M = Utrain.shape
        if self.n_users is None: self.n_users = N
        if self.n_items is None: self.n_items = M
        if (self.n_users, self.n_items) != (N, M):
            raise ValueError("data is %d x %d but the model was declared %d x %d" % (N, M, self.n_users, self.n_items))
        if self.X is not None and (self.X.shape[0] < N or self.Y.shape[0] < M):
            raise ValueError("loaded factors are smaller than the data")
        return self

    def _ensure_factors(self):
        if self.X is not None:
            return
        c = self.config
        ph = int(c["phantom_rows"])
        X0, Y0 = init_factors(self.n_users + ph, self.n_items + ph, c["f"], c["init"], c["seed"])
        self.X = torch.from_numpy(X0.astype(np.float32)).to(self.device)
        self.Y = torch.from_numpy(Y0.astype(np.float32)).to(self.device)

    def _ensure_engine(self):
        if self.engine is not None:
            return
        if self.Utrain is None:
            raise RuntimeError("call input_data() first")
        self._ensure_factors()
        c = self.config
        cu = ops.DeviceCSR.from_scipy(self.Utrain, self.device, self._chunk_nnz())
        ci = cu.transpose()             # on the device (the reference transposes on the host, ALS.py:432)
        self.engine = ALSEngine(cu, ci, self.X, self.Y, c["lamb"], c["alpha"], solver=c["solver"],
                                cg_steps=c["cg_steps"], short_rows=c["short_rows"])

    def _chunk_nnz(self):
        c = self.config["chunk_nnz"]
        return ops.default_chunk_nnz(self.config["f"]) if c is None else int(c)

    # ------------------------------------------------------------------ training
    def train(self, epochs=None, evaluate=None):
        """Runs the remaining epochs (ALSTrainer.fit, ALS.py:343-353).  evaluate: None = evaluate
        after every epoch iff test data was given (the reference's EvalCallback), False = never."""
        if self.mode != "train":
            raise RuntimeError("cannot call train() when mode is not 'train'")        # ALS.py:124
        self._ensure_engine()
        epochs = self.config["epochs"] if epochs is None else int(epochs)
        do_eval = (self.Utest is not None) if evaluate is None else bool(evaluate)
        for epoch in range(self.start_epoch, epochs):
            self.engine.step()
            self._prepared = None                     # the item factor has changed
            self.trace.append(self.engine.last_delta())
            if do_eval:
                self.eval_epochs.append(epoch); self.eval_scores.append(self.evaluate())
            self.start_epoch = epoch + 1
            if self.model_path is not None:
                self.save(self.model_path)                                              # ALS.py:353
        bad = int(self.engine.info.item())
        if bad:
            raise FloatingPointError("%d row systems were not positive definite in float32" % bad)
        return self.trace

    def fit(self, **kwargs):
        """Algorithm.fit (reclibwh/core/Environment.py:26-30; ALSTrainer.fit, ALS.py:320-357): train()."""
        return self.train(**kwargs)

    # ------------------------------------------------------------------ persistence
    def save(self, path=None):
        path = self.model_path if path is None else path
        if path is None:
            raise ValueError("model path not specified")                                # ALS.py:145
        if self.X is None:
            raise RuntimeError("nothing to save")
        os.makedirs(path, exist_ok=True)
        ph = int(self.config["phantom_rows"])
        X = self.X.cpu().numpy().astype(np.float64); Y = self.Y.cpu().numpy().astype(np.float64)
        if ph:
            X, Y = X[:-ph], Y[:-ph]
        np.save(os.path.join(path, "X.npy"), X)
        np.save(os.path.join(path, "Y.npy"), Y)
        c = self.config
        cfg = {"epochs": c["epochs"], "start_epoch": self.start_epoch, "alpha": c["alpha"], "lamb": c["lamb"]}
        with open(os.path.join(path, "algorithm_config.json"), "w") as fp:
            json.dump(cfg, fp)
        if self.eval_scores:
            with open(os.path.join(path, "eval.csv"), "w") as fp:
                fp.write("epoch,score\n")
                for e, s in zip(self.eval_epochs, self.eval_scores):
                    fp.write("%s,%s\n" % (float(e), repr(float(s))))
        if self.trace:
            with open(os.path.join(path, "trace.csv"), "w") as fp:
                fp.write("trace\n" + "\n".join(repr(float(t)) for t in self.trace) + "\n")

    def load(self, path):
        X = np.load(os.path.join(path, "X.npy"), allow_pickle=False)                    # ALS.py:151-154
        Y = np.load(os.path.join(path, "Y.npy"), allow_pickle=False)
        if X.ndim != 2 or Y.ndim != 2 or X.shape[1] != Y.shape[1]:
            raise ValueError("X.npy / Y.npy have inconsistent shapes")
        self.config["f"] = int(X.shape[1])
        self.config["phantom_rows"] = 0
        self.n_users, self.n_items = X.shape[0], Y.shape[0]
        self.X = torch.from_numpy(X.astype(np.float32)).to(self.device)
        self.Y = torch.from_numpy(Y.astype(np.float32)).to(self.device)
        cfgp = os.path.join(path, "algorithm_config.json")
        if os.path.exists(cfgp):
            with open(cfgp) as fp:
                cfg = json.load(fp)
            for k in ("epochs", "alpha", "lamb"):
                if k in cfg: self.config[k] = cfg[k]
            self.start_epoch = int(cfg.get("start_epoch", 0))
        # the epochs already run: a resumed train() appends to them, as the reference's callback keeps its
        # per-epoch scores across epochs and writes them all (EvalProto.py:74-94)
        self.eval_epochs, self.eval_scores, self.trace = [], [], []
        evp, trp = os.path.join(path, "eval.csv"), os.path.join(path, "trace.csv")
        if os.path.exists(evp):
            with open(evp) as fp:
                for line in fp.read().strip().split("\n")[1:]:
                    e, sc = line.split(",")
                    if float(e) < self.start_epoch:
                        self.eval_epochs.append(int(float(e))); self.eval_scores.append(float(sc))
        if os.path.exists(trp):
            with open(trp) as fp:
                self.trace = [float(t) for t in fp.read().strip().split("\n")[1:]][: self.start_epoch]
        self.engine = None
        self._foldin_cache = None
        self._prepared = None
        return self

    # ------------------------------------------------------------------ inference
    def _idx(self, a):
        return torch.as_tensor(np.asarray(a, dtype=np.int64).reshape(-1), device=self.device)

    def predict(self, u_in, i_in):
        """p[k] = <X[u_in[k]], Y[i_in[k]]> for paired index arrays (ALS.py:359-362)."""
        return ops.predict_pairs(self.X, self.Y, self._idx(u_in), self._idx(i_in)).cpu().numpy()

    def recommend(self, users, k=200, filter_rated=False):
        """Ranked items and scores for `users`: (idx [nu, k], scores [nu, k]), best first
        (SimpleMFRecAlgo.recommend, RecAlgos.py:20-32, truncated to the k the reference's consumers
        read: 200).  filter_rated drops items the user rated in the training data
        (RecommenderService.py:16-29)."""
        k = min(int(k), self.n_items)
        ex = None
        if filter_rated and self.Utrain is not None:
            # the training matrix as the engine already holds it on the device; otherwise uploaded once and kept
            if self.engine is not None:
                ex = self.engine.csr_users
            else:
                if self._exclusions is None:
                    self._exclusions = ops.device_exclusions((self.Utrain.indptr, self.Utrain.indices), self.device)
                ex = self._exclusions
        if ex is not None:
            u = np.asarray(users, dtype=np.int64).reshape(-1)
            if u.size and (u.min() < 0 or u.max() >= self.Utrain.shape[0]):
                raise ValueError("filter_rated: user id outside the training matrix (%d users)" % self.Utrain.shape[0])
        idx, val = ops.recommend_topk(self.X, self._items_for_scoring(), self._idx(users), k, M=self.n_items, exclude_csr=ex)
        return idx.cpu().numpy().astype(np.int64), val.cpu().numpy()

    def _items_for_scoring(self):
        """The item factor as recommend() hands it to the top-k: converted once per trained / loaded model into the
        score kernel's operand form (ops.prepare_items: about twice the serving throughput) where the factor count
        allows it, else the factor itself."""
        if not ops.prepared_items_supported(self.Y.shape[1]):
            return self.Y
        if self._prepared is None:
            self._prepared = ops.prepare_items(self.Y, self.n_items)
        return self._prepared

    def similar_to(self, items, k=200):
        """Items most cosine-similar to each of `items` in item-factor space, best first (the item
        itself comes first): SimpleMFSimAlgo.similar with output_emb='Y' (RecAlgos.py:61-73,
        ALS.py:412)."""
        k = min(int(k), self.n_items)
        Yn = ops.normalize_rows(self.Y[: self.n_items])
        idx, val = ops.recommend_topk(Yn, Yn, self._idx(items), k, M=self.n_items)
        return idx.cpu().numpy().astype(np.int64), val.cpu().numpy()

    def similar(self, item, k=200):
        """SimAlgo.similar (Environment.py:20-24; SimpleMFSimAlgo.similar, RecAlgos.py:61-73) for a 1-D array of
        item ids: (ranked ids, cosine similarities), best first, cut to the first k columns."""
        item = np.asarray(item)
        if item.ndim != 1:
            raise ValueError("item must be a 1-D array of item ids")                    # RecAlgos.py:66
        return self.similar_to(item, k)

    def make_update_data(self, data):
        """UpdateAlgo.make_update_data (Environment.py:43-45; ALS.py:387-395): turns (rows, cols, vals)
        triplets into what update_user consumes - here a device CSR over the unique rows, each mapped to its
        factor row, instead of the reference's padded-batch iterator.  No triplets: an empty list, as there."""
        import scipy.sparse as sps
        rows, cols, vals = data
        rows = np.asarray(rows, dtype=np.int64); cols = np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float64)
        if len(rows) == 0:
            return []                                                                   # ALS.py:390
        if not (len(rows) == len(cols) == len(vals)):
            raise ValueError("rows, cols and vals must have the same length")
        if rows.min() < 0 or cols.min() < 0 or rows.max() >= self.X.shape[0] or cols.max() >= self.n_items:
            raise ValueError("row or column id outside the model")
        uniq, inv = np.unique(rows, return_inverse=True)
        U = sps.csr_matrix((vals, (inv, cols)), shape=(len(uniq), self.n_items))   # duplicates sum, as csr_matrix does
        U.sort_indices()
        return ops.DeviceCSR.from_scipy(U, self.device, self._chunk_nnz(), row_ids=uniq)

    def update_user(self, rows, cols=None, vals=None, refresh_gram=True):
        """Fold-in (ALSUpdateAlgo.update_user, ALS.py:375-385): re-solves the factor rows of the given users from
        their triplets alone, against the current item factors.  Call it with (rows, cols, vals) or with the
        object make_update_data returned.

        refresh_gram=True (default) computes the item Gram from the current Y first.  refresh_gram=False
        reproduces the reference: the cached Gram is used as it stands (ALS.py:385 passes the cache 'Y2' and never
        refreshes it) - after train() that is the Gram taken before the last item sweep (ALS.py:345-349), after a
        load it is fresh (ALS.py:301-308), and later fold-ins keep using it."""
        csr = rows if (cols is None and vals is None) else self.make_update_data((rows, cols, vals))
        if isinstance(csr, list) and not csr:
            return
        if not isinstance(csr, ops.DeviceCSR) or csr.row_ids is None:
            raise ValueError("update_user takes (rows, cols, vals) or the result of make_update_data")
        if refresh_gram:
            cache = ops.gram(self.Y, self.config["lamb"])
        elif self.engine is not None and self.trace:
            cache = self.engine.cache_Y                # as the last training epoch left it
        else:                                          # nothing trained in this object: computed once, then reused
            if self._foldin_cache is None:
                self._foldin_cache = ops.gram(self.Y, self.config["lamb"])
            cache = self._foldin_cache
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        f = self.X.shape[1]
        if self.config["solver"] == "cg" or f > 128:
            # a brand-new row starts far from its solution: f steps reach it to rounding
            ops.half_sweep_cg(csr, self.Y, cache, self.X, self.config["alpha"], max(f, self.config["cg_steps"] or 0), info=info)
        else:
            ops.half_sweep(csr, self.Y, cache, self.X, self.config["alpha"], info=info)
        if int(info.item()):
            raise FloatingPointError("fold-in system not positive definite in float32")

    def evaluate(self, users=None):
        """The reference's per-epoch score (AUCEval.evaluate, EvalProto.py:28-62): mean average
        precision over the first 200 recommendations, test items rated above med_score relevant,
        training items filtered.  Default users: those with < 50 training ratings, thinned to
        ~300 (ALS.py:434-437)."""
        if self.Utrain is None or self.Utest is None:
            raise RuntimeError("evaluate() needs training and test data (input_data)")
        users = metrics.users_to_test(self.Utrain) if users is None else np.asarray(users)
        if len(users) == 0:
            return float("nan")
        lim = int(self.config["eval_lim"])
        # on the device end to end: fused top-k, then the average precision from torch operations on the ranked ids
        # (metrics.mean_average_precision is the host loop it is tested against)
        if self._eval_keys is None:
            self._eval_keys = (metrics.pair_keys(self.Utrain, self.n_items, self.device),) + \
                metrics.relevant_test_keys(self.Utest, self.n_items, self.device, self.config["med_score"])
        u = self._idx(users)
        idx, _ = ops.recommend_topk(self.X, self.Y, u, min(lim, self.n_items), M=self.n_items)
        return metrics.mean_average_precision_device(idx, u, *self._eval_keys, self.n_items, lim)

    def rmse(self, U=None):
        """Root mean square error of the predictions over the stored entries of U (default: the test matrix), as
        utils.py:58-59 on the reference's predict (ALS.py:359-362); computed on the device."""
        import scipy.sparse as sps
        U = self.Utest if U is None else sps.csr_matrix(U)
        if U is None or U.nnz == 0:
            return float("nan")
        coo = U.tocoo()
        pred = ops.predict_pairs(self.X, self.Y, self._idx(coo.row), self._idx(coo.col))
        return metrics.rmse_device(pred, torch.as_tensor(coo.data, device=self.device))
