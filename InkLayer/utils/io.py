"""File output helper of this build (no counterpart in the reference, which saves every PNG serially)."""


def save_all(jobs):
    """(PIL image, path) pairs -> files.  PNG encoding is zlib work that releases the GIL: a small thread pool keeps the
    dozens of mask files of a sketch from serialising the runner."""
    jobs = list(jobs)
    if len(jobs) <= 2:
        for im, path in jobs:
            im.save(path)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        list(ex.map(lambda j: j[0].save(j[1]), jobs))
