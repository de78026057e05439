"""PathSeg with the reference's interface (EDaGe-PP/PathSeg.py:9-58), backed by stage A on the GPU."""
import numpy as np

SegLenRange = 7
MinLen = 0


class PathSeg:
    def __init__(self, polyorder=4, dim=2, is_straight=False):
        if polyorder != 4 or dim != 2:
            raise NotImplementedError("the MI355X path implements the generator's fixed geometry: polyorder=4, dim=2")
        self.PolyOrder = polyorder
        self.Poly = np.zeros([polyorder + 1, 1])
        self.EndPoint = 0
        self.Length = 0
        self.Translation = np.zeros([dim, 1])
        self.Rotation = 0
        self.GradSt = 0
        self.GradEnd = 0
        self.is_straight = bool(is_straight)   # the 0.2 coin of the reference is drawn with the samples
        self._forced = bool(is_straight)

    def random(self, poly=None, endpoint=None):
        """Samples one segment (PathSeg.py:21-36). A lone segment is segment 0 of a one-path launch."""
        if poly is not None:
            raise NotImplementedError("explicit polynomials are not part of the accelerated path")
        from Path import Path
        p = Path(seg_num=10, poly_order=4, dim=2, clearance=1, is_straight=self._forced)
        p._first_segment_only = True
        p.generate(show_now=False)
        s = p.PathSeg[0]
        self.__dict__.update(s.__dict__)
        self.Translation = np.array(p.SegPoint[1])     # [EndPoint, p(EndPoint)] before chaining
        return self.Poly, self.EndPoint
