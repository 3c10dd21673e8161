"""Stand-in for the 7 timm symbols /root/reference/models/{cait,swin}.py import
(SURVEY.md Appendix A).  Used ONLY by tests/golden/gen_golden.py in the build
container to import the reference modules; the behaviours restated here are the
2021-era timm ones.  Never imported by the product."""
