from univst_amd.src.sd3.run_content_inversion_sd3 import main, parser  # noqa: F401

if __name__ == "__main__":
    main(parser().parse_args())
