"""GpuColorMIPSearchProcessor -- the caller side of the two kernels.

Counterpart of the reference's
  * `ColorMIPSearchProcessor.findAllColorDepthMatches` (tools cmd/cdsprocess/ColorMIPSearchProcessor.java:8-12,
    LocalColorMIPSearchProcessor.java:55-116, AbstractColorMIPSearchProcessor.findPixelMatch :60-85): every mask against
    every target, keep the pairs for which `ColorMIPSearch.isMatch` holds;
  * `ColorMIPProcessUtils.selectBestMatches` (cmd/cdsprocess/ColorMIPProcessUtils.java:12-34) on top of
    `ItemsHandling.selectTopRankedElements` (API/results/ItemsHandling.java:80-109): top lines (publishedName) ->
    top samples per line (neuronId) -> top matches per sample;
  * `CalculateGradientScoresCmd`: shape scores of the selected pairs (:541-586) and the per-mask normalisation
    (:616-645, GradientAreaGapUtils.calculateNormalizedScore).

The per-pair Java loop becomes: one `cds_search` per mask batch over the resident library, a vectorised match filter,
the (host) top-N selection over the few survivors, one `cds_gradscore` over the selected pairs.  Entities are plain
records here -- persistence (Mongo / JSON writers) stays out of scope.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import cds


@dataclass
class NeuronInfo:
    """The three attributes of AbstractNeuronEntity the selection logic reads (API/model/AbstractNeuronEntity.java)."""
    mipId: str
    publishedName: Optional[str] = None
    neuronId: Optional[str] = None


@dataclass
class CDMatch:
    """The score fields of CDMatchEntity (API/model/CDMatchEntity.java:12-169)."""
    maskIndex: int
    targetIndex: int
    matchingPixels: int
    matchingPixelsRatio: float
    mirrored: bool
    gradientAreaGap: Optional[int] = None
    highExpressionArea: Optional[int] = None
    normalizedScore: Optional[float] = None
    errors: Optional[str] = None

    def hasGradScore(self):
        return (self.gradientAreaGap is not None and self.gradientAreaGap >= 0 and
                self.highExpressionArea is not None and self.highExpressionArea >= 0)

    def getGradScore(self):  # CDMatchEntity.java:75-84
        return cds.calculate2DShapeScore(self.gradientAreaGap, self.highExpressionArea) if self.hasGradScore() else -1


def selectTopRankedElements(items: Sequence, groupingCriteria, scoreExtractor, topResults: int, limitSubResults: int):
    """ItemsHandling.selectTopRankedElements (API/results/ItemsHandling.java:80-109): group, sort each group by
    descending score (stable), optionally cut each group, rank the groups by their best score (stable), cut."""
    groups = {}
    for it in items:
        key = groupingCriteria(it)
        if key is None or str(key).strip() == "":
            key = "UNKNOWN"
        groups.setdefault(key, []).append(it)
    ranked = []
    for key, lst in groups.items():
        lst = sorted(lst, key=lambda v: -float(scoreExtractor(v)))
        if 0 < limitSubResults < len(lst):
            lst = lst[:limitSubResults]
        ranked.append((key, max(float(scoreExtractor(v)) for v in lst), lst))
    ranked.sort(key=lambda e: -e[1])
    if 0 < topResults < len(ranked):
        ranked = ranked[:topResults]
    return ranked


class GpuColorMIPSearchProcessor:
    """masks: uint8[n_masks,H,W,3]; the targets live in a `cds.Library` (+ `targets` metadata, one NeuronInfo per slot)."""

    def __init__(self, colorMIPSearch: cds.ColorMIPSearch, library: cds.Library, targets: Optional[List[NeuronInfo]] = None,
                 first=0, count=None):
        self.colorMIPSearch = colorMIPSearch
        self.library = library
        self.first = int(first)
        self.count = library.capacity - self.first if count is None else int(count)
        self.targets = targets or [NeuronInfo(str(i), f"line{i}", f"sample{i}") for i in range(self.count)]
        assert len(self.targets) == self.count

    # ---- colorDepthSearch -------------------------------------------------------------------------
    def findAllColorDepthMatches(self, masks) -> List[CDMatch]:
        alg = self.colorMIPSearch.createQueryColorDepthSearchWithDefaultThreshold(masks)
        scores, mirrored = alg.searchLibrary(self.library, self.first, self.count)
        sizes = alg.getQuerySizes()
        out = []
        for m in range(alg.n_masks):
            if sizes[m] == 0:  # "No computation created ... because it is empty" (LocalColorMIPSearchProcessor.java:88-92)
                continue
            keep = self.colorMIPSearch.isMatchArray(scores[m], int(sizes[m]))
            ratio = (scores[m].astype(np.float64) / float(sizes[m])).astype(np.float32)
            for t in np.flatnonzero(keep):
                out.append(CDMatch(m, int(t), int(scores[m, t]), float(ratio[t]), bool(mirrored[m, t])))
        return out

    # ---- gradientScores ---------------------------------------------------------------------------
    def selectBestMatches(self, matches: List[CDMatch], topLineMatches, topSamplesPerLine, topMatchesPerSample) -> List[CDMatch]:
        """Per mask, as CalculateGradientScoresCmd.getCDMatchesForMaskMipIDs does (:323-379)."""
        by_mask = {}
        for mt in matches:
            by_mask.setdefault(mt.maskIndex, []).append(mt)
        out = []
        for m in sorted(by_mask):
            lines = selectTopRankedElements(by_mask[m], lambda v: self.targets[v.targetIndex].publishedName,
                                            lambda v: v.matchingPixels, topLineMatches, -1)
            for _, _, line_items in lines:
                samples = selectTopRankedElements(line_items, lambda v: self.targets[v.targetIndex].neuronId,
                                                  lambda v: v.matchingPixels, topSamplesPerLine, topMatchesPerSample)
                for _, _, sample_items in samples:
                    out.extend(sample_items)
        return out

    def calculateGradientScores(self, masks, matches: List[CDMatch], queryThreshold, mirrorMask=True, roiMask=None,
                                excludedRegions=cds.label_regions) -> List[CDMatch]:
        """Shape scores for `matches` (one launch), then the per-mask normalisation."""
        if not matches:
            return matches
        alg = cds.Shape2DMatchColorDepthSearchAlgorithm(masks, queryThreshold, mirrorMask, roiMask, excludedRegions,
                                                        context=self.library.ctx)
        pm = np.array([mt.maskIndex for mt in matches], dtype=np.int32)
        ps = np.array([self.first + mt.targetIndex for mt in matches], dtype=np.int64)
        gap, high, mir = alg.scoreLibraryPairs(self.library, pm, ps)
        for mt, g, h in zip(matches, gap, high):
            # "only set the scores that were actually calculated" (CalculateGradientScoresCmd.java:564-570); the
            # match keeps the mirrored flag of the pixel-match scan
            if g != -1:
                mt.gradientAreaGap = int(g)
            if h != -1:
                mt.highExpressionArea = int(h)
        self.normalizeScores(matches)
        return matches

    @staticmethod
    def normalizeScores(matches: List[CDMatch]):
        """CalculateGradientScoresCmd.normalizeScores (:616-645)."""
        by_mask = {}
        for mt in matches:
            by_mask.setdefault(mt.maskIndex, []).append(mt)
        for items in by_mask.values():
            max_pix = max([-1] + [mt.matchingPixels for mt in items])
            max_grad = max([-1] + [mt.getGradScore() for mt in items])
            for mt in items:
                mt.normalizedScore = float(np.float32(cds.calculateNormalizedScore(mt.matchingPixels, mt.getGradScore(), max_pix, max_grad)))
