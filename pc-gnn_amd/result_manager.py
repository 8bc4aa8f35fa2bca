"""Experiment bookkeeping in the reference's formats (src/result_manager.py:18-157), so that the tooling around the reference
- the notebooks that aggregate ``experimental_results/*`` and ``eval.ipynb``'s best-model lookup - reads runs of this
framework unchanged:

    ./experimental_results/validation_log/{exp_id}.log   "key: value" lines of the config, then per validation
                                                         "[Epoch-007] Validation performance" + the metric line (utils.py:325)
    ./experimental_results/test_log/{exp_id}.log         the config lines, then "Test performance: - Epoch_Best: E\\t" + metric line
    ./experimental_results/validation_df/{exp_id}.pkl    DataFrame: epoch, epoch_best, accuracy, f1, f1_macro, precision,
                                                         precision_macro, recall, recall_macro, auc
    ./experimental_results/test_df/{model}-{data}.pkl    DataFrame of every finished run of the pair: exp_id, epoch_best, the
                                                         eight metrics, every config key (rebuilt from the test logs first)
    ./experimental_results/saved_models/{exp_id}.pickle  torch.save(model.state_dict())
    ./experimental_results/predictions/{exp_id}-{name}.npy

``exp_id = {model}-{data_name}-{yymmdd-HHMMSS-ffffff}`` (result_manager.py:37).  Host-only code: nothing here touches the GPU.
"""
import os
from datetime import datetime
from typing import Dict, Optional

import numpy as np

EXP_RES_DIR = "./experimental_results"
METRICS = ("accuracy", "f1", "f1_macro", "precision", "precision_macro", "recall", "recall_macro", "auc")
# metric name in a log line (utils.py:325, lower-cased) -> DataFrame column (result_manager.py:66-73)
_LOG_TO_COLUMN = {"accuracy": "accuracy", "f1": "f1", "f1-macro": "f1_macro", "precision": "precision", "ap": "precision_macro",
                  "recall": "recall", "recall-macro": "recall_macro", "auc-roc": "auc"}


def _subdir(root: str, name: str) -> str:
    path = f"{root}/{name}"
    os.makedirs(path, exist_ok=True)
    return path


class ResultManager:
    def __init__(self, args: Dict, root: str = EXP_RES_DIR) -> None:
        import pandas as pd
        self._pd = pd
        self.args = args
        os.makedirs(root, exist_ok=True)
        self.model_dir, self.pred_dir = _subdir(root, "saved_models"), _subdir(root, "predictions")
        df_val_dir, df_test_dir = _subdir(root, "validation_df"), _subdir(root, "test_df")
        log_val_dir, self.log_test_dir = _subdir(root, "validation_log"), _subdir(root, "test_log")
        model, data_name = args["model"], args["data_name"]
        self.exp_id = f"{model}-{data_name}-{datetime.now().strftime('%y%m%d-%H%M%S-%f')}"          # :37
        self.df_val_path = os.path.join(df_val_dir, f"{self.exp_id}.pkl")
        self.df_test_path = os.path.join(df_test_dir, f"{model}-{data_name}.pkl")
        self.log_val_path = os.path.join(log_val_dir, f"{self.exp_id}.log")
        self.log_test_path = os.path.join(self.log_test_dir, f"{self.exp_id}.log")
        self.model_path = os.path.join(self.model_dir, f"{self.exp_id}.pickle")
        self.df_val = pd.DataFrame()
        self.df_test = pd.read_pickle(self.df_test_path) if os.path.exists(self.df_test_path) else pd.DataFrame()
        header = self.get_configuration_line()[1:]                                                # :84-89
        for path in (self.log_val_path, self.log_test_path):
            with open(path, "a") as f:
                f.write(header + "\n")

    def get_configuration_line(self) -> str:
        return "".join(f"\n{key}: {self.args[key]}" for key in sorted(self.args.keys()))        # :77-81

    # -- logs + frames ---------------------------------------------------------------------------------------------
    def _metric_row(self, df, idx, values):
        for col, v in zip(METRICS, values):
            df.loc[idx, col] = v

    def write_val_log(self, epoch: int, epoch_best: int, accuracy: float, f1: float, f1_macro: float, precision: float,
                      precision_macro: float, recall: float, recall_macro: float, auc: float, line: str,
                      print_line: bool = True) -> None:
        line = f"[Epoch-{str(epoch).zfill(3)}] Validation performance\n{line}"                    # :97
        with open(self.log_val_path, "a") as f:
            f.write(line + "\n")
        if print_line:
            print(line)
        idx = len(self.df_val)
        self.df_val.loc[idx, "epoch"] = epoch
        self.df_val.loc[idx, "epoch_best"] = epoch_best
        self._metric_row(self.df_val, idx, (accuracy, f1, f1_macro, precision, precision_macro, recall, recall_macro, auc))
        self.df_val.to_pickle(self.df_val_path)

    def load_df_test(self) -> None:
        """Rebuild the pair's test frame from every finished run's test log (result_manager.py:47-75)."""
        pd = self._pd
        df = pd.DataFrame()
        pair = f"{self.args['model']}-{self.args['data_name']}"
        for filename in os.listdir(self.log_test_dir):
            if pair not in filename:
                continue
            with open(os.path.join(self.log_test_dir, filename)) as f:
                lines = [ln.strip() for ln in f.readlines()][:-1]          # (the metric line ends with "\t\n" + "\n": drop the empty one)
            if not lines:
                continue
            result = lines.pop()
            if "Test performance" not in result:
                continue
            idx = len(df)
            df.loc[idx, "exp_id"] = filename[:-4]
            parsed = dict(tuple(m.strip().split(": ")) for m in result.split("- ")[1:])
            parsed = {k.lower(): float(v) for k, v in parsed.items()}
            df.loc[idx, "epoch_best"] = parsed["epoch_best"]
            for key, col in _LOG_TO_COLUMN.items():
                df.loc[idx, col] = parsed[key]
            cfg = dict(tuple(ln.split(": ")) for ln in lines)
            for key in sorted(cfg.keys()):
                df.loc[idx, key] = cfg[key]
        df.to_pickle(self.df_test_path)
        self.df_test = df

    def write_test_log(self, epoch_best: int, accuracy: float, f1: float, f1_macro: float, precision: float,
                       precision_macro: float, recall: float, recall_macro: float, auc: float, line: str,
                       print_line: bool = True) -> None:
        self.load_df_test()
        line = f"Test performance: - Epoch_Best: {epoch_best}\t" + line                           # :121
        with open(self.log_test_path, "a") as f:
            f.write(line + "\n")
        if print_line:
            print(line)
        idx = len(self.df_test)
        self.df_test.loc[idx, "exp_id"] = self.exp_id
        self.df_test.loc[idx, "epoch_best"] = epoch_best
        self._metric_row(self.df_test, idx, (accuracy, f1, f1_macro, precision, precision_macro, recall, recall_macro, auc))
        for key in sorted(self.args.keys()):
            self.df_test.loc[idx, key] = self.args[key]
        self.df_test.to_pickle(self.df_test_path)

    # -- lookups ---------------------------------------------------------------------------------------------------------
    def get_best_model_exp_id(self, metric: Optional[str] = "auc") -> str:
        return self.df_test.iloc[self.df_test[metric].argmax()]["exp_id"]                         # :145

    def get_best_model_path(self, metric: Optional[str] = "auc") -> str:
        return os.path.join(self.model_dir, f"{self.get_best_model_exp_id(metric)}.pickle")      # :153

    def save_predictions(self, arr: np.ndarray, name: str) -> None:
        np.save(os.path.join(self.pred_dir, f"{self.exp_id}-{name}"), arr)                        # :156
