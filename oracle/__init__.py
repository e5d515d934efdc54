"""CPU oracle for the ExCEL training-free CAM + affinity + PAR hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a plain-numpy (fp32) restatement of the reference algorithm
(zwyang6/ExCEL, snapshot 2025-09-05), each function citing the reference
file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it -- and there only as the
checker / the timed CPU baseline, never as part of the shipped data path.
``excel_amd`` (the product) never imports ``oracle``.

Pinning status (see DESIGN.md "Oracle"):
  * pinned against goldens captured by importing the reference's own Python
    in the build container (tests/golden/make_goldens.py -> tests/golden/*.npz):
      ViT surgery forward, generate_clip_fts, clip_feature_surgery,
      attr_aggregate, compute_trans_mat, refine_cams_with_aff (cv2 stubbed),
      refine_cams_with_bkg_weclip (cv2 stubbed), PAR, evaluate.scores.
  * PARITY UNPINNED for the two OpenCV calls the reference makes
    (cv2.findContours/boundingRect in utils/affutils.py:33-47 and cv2.resize in
    utils/affutils.py:75): OpenCV is a third-party dependency that is absent
    from /root/reference and from this image (version unpinned upstream,
    requirements.txt does not list it).  Their published semantics are
    restated in oracle/aff.py and covered by hand-made known-answer cases.
  * PARITY UNPINNED for the DenseCRF stage (oracle/dcrf.py): pydensecrf (conda pin
    1.0rc3, requirements.txt:4) is third-party, absent from /root/reference and
    from this image, and the reference holds no vector for it; the published
    mean-field / permutohedral-lattice algorithm is restated and covered by
    property tests (constant preservation, symmetry, partition of unity).
"""

from . import interp, vit, cam, attr, aff, par, evaluate, pipeline, decoder, text, dcrf  # noqa: F401
