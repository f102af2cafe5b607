from univst_amd.src.sd.run_style_inversion_sd import main, parser
if __name__ == "__main__":
    main(parser().parse_args())
