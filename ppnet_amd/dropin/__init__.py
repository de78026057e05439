"""Drop-in modules with the reference's own names and signatures (EDaGe-PP/PathSeg.py, Path.py,
PathGenerate.py, MapGenerate.py, process_map.py).  Put this directory first on sys.path and the
reference's scripts (`from Path import Path`, `from PathGenerate import PathGroup`, ...) run on the HIP path.
"""
