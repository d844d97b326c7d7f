"""Rating-file ingest for RecommenderALS.input_data(): the part of the reference's
ExplicitDataFromCSV that the ALS path touches (reclibwh/utils/ItemMetadata.py:267-405,
reclibwh/utils/utils.py:8-33).

  from a raw CSV      user / item keys become dense ids in sorted-key order (pandas category codes,
                      utils.py:19-33: normalizeDf), ratings float64, an 80/20 train/test split
  from a saved folder ratings_sanitized.csv (rating,item,user) + ratings_split.csv (split) +
                      user_map.csv + item_map.csv, the files ExplicitDataFromCSV.save writes
                      (ItemMetadata.py:367-374); metadata.csv / stats.csv are not needed by ALS
  .N / .M             number of users / items (ItemMetadata.py:325-331)
  .make_training_datasets(dtype='sparse') -> (Utrain, Utest) scipy CSR float64 [N, M]
                      (ItemMetadata.py:378-405; duplicate (user, item) pairs are summed, as
                      scipy's csr_matrix((data, (rows, cols))) does there)

Host-side, pandas only: it runs once per data set, outside the timed path.  The split of the
reference is unseeded; here it takes a seed so that runs are repeatable.
"""
import os

import numpy as np


class ExplicitDataFromCSV:
    def __init__(self, from_saved=False, **kwargs):
        if from_saved:
            self._from_folder(**kwargs)
        else:
            self._from_raw_csv(**kwargs)

    # ------------------------------------------------------------------ construction
    def _from_raw_csv(self, ratings_csv, r_item_col="movieId", r_user_col="userId", r_rating_col="rating",
                      train_test_split=0.8, seed=0, **_ignored):
        import pandas as pd
        df = pd.read_csv(ratings_csv, usecols=[r_item_col, r_user_col, r_rating_col])
        self.user_keys, user = np.unique(df[r_user_col].to_numpy(), return_inverse=True)   # sorted keys = category codes
        self.item_keys, item = np.unique(df[r_item_col].to_numpy(), return_inverse=True)
        self.user = user.astype(np.int64); self.item = item.astype(np.int64)
        self.rating = df[r_rating_col].to_numpy().astype(np.float64)
        if not (0.0 < train_test_split <= 1.0):
            raise ValueError("train_test_split must be in (0, 1]")
        perm = np.random.default_rng(seed).permutation(len(self.rating))
        self.split = np.zeros(len(self.rating), dtype=np.int64)
        self.split[perm[int(train_test_split * len(perm)):]] = 1

    def _from_folder(self, data_folder, load_ratings=True, **_ignored):
        import pandas as pd
        umap = pd.read_csv(os.path.join(data_folder, "user_map.csv"))
        imap = pd.read_csv(os.path.join(data_folder, "item_map.csv"))
        self.user_keys = umap.sort_values("user_cat")["user"].to_numpy()
        self.item_keys = imap.sort_values("item_cat")["item"].to_numpy()
        if load_ratings:
            r = pd.read_csv(os.path.join(data_folder, "ratings_sanitized.csv"), index_col=None)[["rating", "item", "user"]]
            s = pd.read_csv(os.path.join(data_folder, "ratings_split.csv"), index_col=None)
            if len(r) != len(s):
                raise ValueError("ratings_sanitized.csv and ratings_split.csv differ in length")   # ItemMetadata.py:283
            self.user = r["user"].to_numpy().astype(np.int64); self.item = r["item"].to_numpy().astype(np.int64)
            self.rating = r["rating"].to_numpy().astype(np.float64)
            self.split = s.iloc[:, 0].to_numpy().astype(np.int64)
            if self.user.size and (self.user.min() < 0 or self.user.max() >= self.N or self.item.min() < 0
                                   or self.item.max() >= self.M):
                raise ValueError("rating ids outside the user / item maps")
        else:
            self.user = self.item = self.split = np.zeros(0, dtype=np.int64)
            self.rating = np.zeros(0, dtype=np.float64)

    # ------------------------------------------------------------------ the contract input_data() uses
    @property
    def N(self):
        return len(self.user_keys)

    @property
    def M(self):
        return len(self.item_keys)

    @property
    def Nranked(self):
        return len(self.rating)

    def make_training_datasets(self, dtype="sparse", ctype=np.int32, rtype=np.float64, users=None):
        import scipy.sparse as sps
        if dtype != "sparse":
            raise ValueError("only dtype='sparse' is provided (the ALS path uses nothing else, ALS.py:427)")
        keep = np.ones(len(self.rating), dtype=bool) if users is None else np.isin(self.user, np.asarray(list(users)))

        def mat(which):
            m = keep & (self.split == which)
            U = sps.csr_matrix((self.rating[m].astype(rtype), (self.user[m].astype(ctype), self.item[m].astype(ctype))),
                               shape=(self.N, self.M))
            U.sort_indices()
            return U

        return mat(0), mat(1)

    def save(self, dir):
        import pandas as pd
        os.makedirs(dir, exist_ok=True)
        pd.DataFrame({"rating": self.rating, "item": self.item, "user": self.user}).to_csv(
            os.path.join(dir, "ratings_sanitized.csv"), index=False)
        pd.DataFrame({"split": self.split}).to_csv(os.path.join(dir, "ratings_split.csv"), index=False)
        pd.DataFrame({"user": self.user_keys, "user_cat": np.arange(self.N)}).to_csv(
            os.path.join(dir, "user_map.csv"), index=False)
        pd.DataFrame({"item": self.item_keys, "item_cat": np.arange(self.M)}).to_csv(
            os.path.join(dir, "item_map.csv"), index=False)
