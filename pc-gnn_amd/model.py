"""PC-GNN model head over the HIP hot path (reference: src/model.py:13-62)."""
import torch
import torch.nn as nn
from torch.nn import init


class PCALayer(nn.Module):
    """One Pick-Choose-Aggregate layer: ``W_cls . embeds`` -> 2 logits, the two-term
    loss ``xent(gnn) + lambda_1 * xent(label_aware)`` (model.py:34-62)."""

    def __init__(self, num_classes, inter1, lambda_1):
        super().__init__()
        self.inter1 = inter1
        self.xent = nn.CrossEntropyLoss()
        self.weight = nn.Parameter(torch.FloatTensor(num_classes, inter1.embed_dim))
        init.xavier_uniform_(self.weight)
        self.lambda_1 = lambda_1
        self.epsilon = 0.1

    def forward(self, nodes, labels, train_flag=True):
        embeds1, label_scores = self.inter1(nodes, labels, train_flag)   # [E,B], [B,2]
        return self.weight.mm(embeds1).t(), label_scores                 # model.py:38-39

    def to_prob(self, nodes, labels, train_flag=True):
        gnn_logits, label_logits = self.forward(nodes, labels, train_flag)
        return torch.sigmoid(gnn_logits), torch.sigmoid(label_logits)    # model.py:43-45

    def loss(self, nodes, labels, train_flag=True):
        gnn_scores, label_scores = self.forward(nodes, labels, train_flag)
        y = torch.as_tensor(labels, device=gnn_scores.device).long().reshape(-1)
        label_loss = self.xent(label_scores, y)                          # Eq. (7)
        gnn_loss = self.xent(gnn_scores, y)                              # Eq. (10)
        return gnn_loss + self.lambda_1 * label_loss                     # Eq. (11)

    def check(self):
        """Raise if a batch overflowed its selection list since the last check (synchronises; not in the reference:
        its Python sets cannot overflow).  ``utils.test`` calls it once per evaluation pass."""
        if hasattr(self.inter1, "check"):
            self.inter1.check()
